/* x264hip.h -- C ABI of the MI355X (gfx950) lookahead / motion-estimation path.
 *
 * Drop-in boundary for the hot path named in BASELINE.json: every entry point states the reference
 * interface (file:line under jpsdr/x264) it replaces.  Plain pointers and sizes only; no C++ or
 * torch types.  All functions return 0 on success and a negative X264HIP_E* code on failure, never
 * throw, and never fall back to a CPU implementation: without a usable HIP device x264hip_open()
 * fails with X264HIP_ENODEV.
 *
 * Threading: one context is driven by one thread at a time (like the reference's single lookahead thread,
 * encoder/lookahead.c:90-128).  Different contexts are independent and may be driven concurrently from different
 * threads, on the same or on different devices (every call selects its context's device for the calling thread);
 * two contexts on independent GOP segments are how one device is kept busy while a segment's decisions run on the host.
 * The caller owns every host buffer; the context owns device memory.
 *
 * Bit depth: pixels are uint8_t (bit_depth 8) or uint16_t (bit_depth 10) exactly like the
 * reference's `pixel` (common/common.h:93-105); strides are in pixels.
 */
#ifndef X264HIP_H
#define X264HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X264HIP_OK        0
#define X264HIP_ENODEV   -1   /* no HIP device / runtime error at open */
#define X264HIP_EINVAL   -2   /* bad argument */
#define X264HIP_ENOMEM   -3
#define X264HIP_EDEVICE  -4   /* a kernel or copy failed; the context is latched broken (cf. slicetype-cl.c:44-56) */
#define X264HIP_ETIMEOUT -5   /* in-kernel dependency wait exceeded its bound */
#define X264HIP_ESTATE   -6   /* call sequence error (e.g. evaluation of a frame that was never put) */

#define X264HIP_ME_DIA 0
#define X264HIP_ME_HEX 1
#define X264HIP_BFRAME_MAX 16 /* X264_BFRAME_MAX, common/base.h */

typedef struct x264hip_ctx x264hip_ctx;

/* What lowres_context_init() (encoder/slicetype.c:45-61) and x264_param_t give the reference's
 * lookahead.  cost_mv is the centred table h->cost_mv[X264_LOOKAHEAD_QP] (encoder/analyse.c:151-157),
 * valid for indices [-2*4*mv_range, +2*4*mv_range]; it is copied at open. */
#define X264HIP_LOOKAHEAD_SLICES_MAX 16 /* X264_LOOKAHEAD_THREAD_MAX, common/common.h */
typedef struct x264hip_params
{
    int bit_depth;        /* 8 or 10 */
    int width, height;    /* picture size as given to the encoder (any size; mod16 handling is internal) */
    int bframes;          /* param.i_bframe */
    int lambda;           /* x264_lambda_tab[X264_LOOKAHEAD_QP] */
    int me_method;        /* X264HIP_ME_DIA / X264HIP_ME_HEX: lookahead h->mb.i_me_method */
    int subpel_refine;    /* lookahead h->mb.i_subpel_refine: 2 or 4 */
    int me_range;         /* param.analyse.i_me_range */
    int mv_range;         /* param.analyse.i_mv_range */
    int subme;            /* param.analyse.i_subpel_refine */
    int mbcmp_satd;       /* h->pixf.mbcmp == satd (encoder.c:1409-1427) */
    int fpelcmp_satd;     /* h->pixf.fpelcmp == satd (me=tesa) */
    int weighted_bipred;  /* param.analyse.b_weighted_bipred */
    int aq_mode;          /* param.rc.i_aq_mode: 0 none, 1 variance, 2 auto-variance, 3 auto-variance biased */
    float aq_strength;    /* param.rc.f_aq_strength */
    int bframe_bias;      /* param.i_bframe_bias */
    int max_frames;       /* frame slots to keep resident (>= lookahead depth + bframes + 3) */
    int no_edges;         /* 1 = slicetype_slice_cost's do_edges == 0 (encoder/slicetype.c:823-828: no MB-tree and no VBV): the
                           * outermost ring of blocks is never evaluated, its vectors stay zero and its intra costs 0xFFFF;
                           * ignored (edges evaluated) for frames of at most 2 blocks in a direction.  0 = every block */
    int lookahead_slices; /* param.i_lookahead_threads (encoder/encoder.c:1273-1300): the frame is searched in that many
                           * horizontal bands, rows [(mb_h*i + n/2)/n, (mb_h*(i+1) + n/2)/n), and a band does not use the
                           * vectors of the band below as predictors (slicetype.c:668,917-918).  0 or 1 = one band */
    int chroma_format;    /* of the Cb / Cr planes handed to frame_put: 0 or 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 (CHROMA_420/422/444; only
                           * adaptive quantisation looks at chroma: block size and shift of ac_energy_plane, ratecontrol.c:238-256) */
    const uint16_t *cost_mv;
} x264hip_params;

typedef struct x264hip_weight
{
    int on, scale, denom, offset; /* x264_weight_t i_scale/i_denom/i_offset, weightfn != NULL (common/mc.h:235-245) */
} x264hip_weight;

/* per-evaluation outputs of slicetype_frame_cost (encoder/slicetype.c:946-991), before the
 * B-frame *100/(120+bias) scaling which stays with the caller */
typedef struct x264hip_cost
{
    int cost_est, cost_est_aq, intra_mbs;
    int intra_cost_est, intra_cost_est_aq; /* the [0][0] cell; defined only when the call was made with with_intra != 0 */
} x264hip_cost;

/* MB-tree step list (encoder/slicetype.c:1051-1184): the host decides the order, the backend does the per-MB work */
#define X264HIP_MBT_ZERO      0   /* memset( frames[slot_b]->i_propagate_cost, 0 ) */
#define X264HIP_MBT_PROPAGATE 1   /* macroblock_tree_propagate( p0, p1, b, referenced ) */
#define X264HIP_MBT_FINISH    2   /* macroblock_tree_finish( frames[slot_b] ) -> f_qp_offset */
#define X264HIP_MBT_SWAP      3   /* XCHG( frames[slot_b]->i_propagate_cost, frames[slot_p0]->i_propagate_cost ): lookahead-less MB-tree */
#define X264HIP_MBT_RESET_QP  4   /* memcpy( frames[slot_b]->f_qp_offset, f_qp_offset_aq ) (slicetype.c:1119-1120) */
typedef struct x264hip_mbtree_op
{
    int type;
    int slot_b, slot_p0, slot_p1;
    int dist_p0, dist_p1;     /* b-p0, p1-b */
    int referenced;
    int bipred_weight;        /* list-0 weight; list 1 uses 64 - bipred_weight */
    float fps_factor;         /* PROPAGATE: float factor (slicetype.c:1063) */
    int fps_factor_i;         /* FINISH: integer factor (slicetype.c:1031) */
    float weightdelta, strength;
} x264hip_mbtree_op;

/* ---- context ---------------------------------------------------------------------------------
 * Replaces x264_opencl_lookahead_init / _delete (encoder/encoder.c:1744-1753, 4208-4209,
 * common/opencl.c:411) as the device bring-up of the coarse lookahead hook.
 * The context works on non-blocking streams of its own; x264hip_open returns when its buffers are initialised (it waits for the
 * device's NULL stream, where those fills are ordered -- and with them for whatever the caller has queued there). */
int  x264hip_open( x264hip_ctx **out, int device, const x264hip_params *params );
void x264hip_close( x264hip_ctx *ctx );
const char *x264hip_strerror( int code );
int  x264hip_device_name( x264hip_ctx *ctx, char *buf, size_t cap );
int  x264hip_synchronize( x264hip_ctx *ctx );  /* x264_opencl_flush (encoder/slicetype-cl.c:58-80) */
/* Hands everything the context has queued on the host to the device without waiting for it: x264hip_mbtree keeps the step lists of
 * consecutive macroblock_tree() calls back so that one launch runs several of them side by side; every call that reads what they
 * write flushes them itself, this entry is for a caller that will not make such a call (end of a stream whose quantiser offsets
 * nobody fetches). */
int  x264hip_flush( x264hip_ctx *ctx );

/* ---- frame ingest ----------------------------------------------------------------------------
 * Replaces, per input frame: x264_adaptive_quant_frame (encoder/ratecontrol.c:304-415, aq-mode 0/1),
 * x264_frame_init_lowres + frame_init_lowres_core + x264_frame_expand_border_lowres
 * (common/mc.c:458-507, common/frame.c:627-631) and x264_opencl_lowres_init
 * (encoder/slicetype-cl.c:129-282).  slot is the caller's frame handle in [0, max_frames).
 * luma may be a host or a device pointer (is_device).  cb/cr are optional 4:2:0 planes used only by
 * AQ; inv_qscale (mb_w*mb_h, Q8) overrides the AQ result when not NULL.
 * A device picture is read on the context's own (non-blocking) stream, which waits for no stream of the caller's: the picture must be
 * complete when the call is made, and stay untouched until the frame's ingest has run (x264hip_synchronize, or any result of the frame).
 * Resets the slot's search/cost state like mc.c:471-481. */
int  x264hip_frame_put( x264hip_ctx *ctx, int slot, const void *luma, int stride, int is_device,
                        const void *cb, const void *cr, int cstride, const uint16_t *inv_qscale );
/* The same for n frames (all with the same stride): one launch per ingest kernel instead of one per frame.  The pointers are device
 * pointers or -- all of them, luma only -- HOST pointers, which is what x264_encoder_encode is handed (encoder/encoder.c:3368-3454,
 * common/frame.c:445-447): the pictures then travel on the device's transfer queue (one per device, shared by its contexts: a call's
 * transfers follow each other at the link's full rate), sixteen to a group; a context alone on its device runs each group's ingest
 * kernels behind its copies while the next group is on its way, beside other contexts they are launched once behind the call's last
 * transfer; nothing waits for the compute stream, and nothing for the device under the queue's lock.  Buffers the DMA engines can
 * read where they are (hipHostMalloc / hipHostRegister: the caller's choice) are copied from there, pageable ones through a ring of
 * pinned staging copies (a memcpy per picture).  x264hip_frame_put with is_device == 0 takes the same road for one picture.
 * x264hip_host_transfer_stats: bytes copied so far, pictures taken as they were / through the ring. */
int  x264hip_frame_put_batch( x264hip_ctx *ctx, int n, const int *slots, const void *const *luma_dev, int stride );
int  x264hip_host_transfer_stats( x264hip_ctx *ctx, uint64_t out[3] );
/* how they travelled: out[0] = transfers that carried a whole group of pictures (pictures that follow each other in pinned memory cross
 * PCIe as ONE transfer per group of sixteen), out[1] = single pictures fetched by a copy kernel on the compute stream (x264hip_frame_put:
 * no DMA stream and no cross-stream event in front of the ingest kernels an encoder-paced caller waits for) */
int  x264hip_host_transfer_stats2( x264hip_ctx *ctx, uint64_t out[2] );
/* same with the 4:2:0 chroma planes of every frame (device pointers; both arrays NULL = luma only) */
int  x264hip_frame_put_batch_yuv( x264hip_ctx *ctx, int n, const int *slots, const void *const *luma_dev, int stride,
                                  const void *const *cb_dev, const void *const *cr_dev, int cstride );
/* i_pixel_sum[0] / i_pixel_ssd[0] of the frame (ratecontrol.c:225-234,405-414) */
int  x264hip_frame_stats( x264hip_ctx *ctx, int slot, uint64_t *pixel_sum, uint64_t *pixel_ssd );

/* ---- evaluation ------------------------------------------------------------------------------
 * x264hip_frame_cost replaces the body of slicetype_frame_cost after its memo check
 * (encoder/slicetype.c:869-991: the slicetype_slice_cost / slicetype_mb_cost loops and the
 * accumulator sums) == x264_opencl_motionsearch + x264_opencl_finalize_cost
 * (encoder/slicetype-cl.c:407-649).
 *   slot_p0/p1/b : frame handles; p0 == p1 == b is the intra-only evaluation.
 *   dist_p0 = b-p0, dist_p1 = p1-b in frames.
 *   do_search[l] : run the motion search of list l now (the caller mirrors the reference's
 *                  lowres_mvs[l][d][0][0] == 0x7FFF first-trigger test, slicetype.c:855-867).
 *   w            : luma weight for the list-0 search when this call triggers it as a P frame, or NULL.
 *   with_intra   : !fenc->b_intra_calculated.
 *   ref1_l0_valid: frames[p1]->lowres_mvs[0][p1-p0-1] has been searched (slicetype.c:629).
 * Results stay resident (mvs, mv costs, lowres_costs, intra costs, row satds) and are fetched with
 * the getters below, which is what the reference's deferred memcpys do (slicetype-cl.c:254-282). */
int  x264hip_frame_cost( x264hip_ctx *ctx, int slot_p0, int slot_p1, int slot_b, int dist_p0, int dist_p1,
                         const int do_search[2], const x264hip_weight *w, int with_intra, int ref1_l0_valid,
                         x264hip_cost *out );

/* weight_cost_luma without the slice-header term (encoder/slicetype.c:191-222): sum over blocks of
 * min( mbcmp( weight(ref lowres block), fenc lowres block ), intra_cost ).  w may be NULL. */
int  x264hip_weight_cost( x264hip_ctx *ctx, int slot_fenc, int slot_ref, const x264hip_weight *w, unsigned *cost );
/* Speculative form: enqueue the unweighted and the weighted cost of n (fenc, ref, weight) triples in one launch and
 * return at once.  A later x264hip_weight_cost() for the same frames (and the same weight, or NULL) is answered
 * from these results without a launch; anything else is computed on demand as before.  Never changes results. */
int  x264hip_prefetch_weight_costs( x264hip_ctx *ctx, int n, const int *slot_fenc, const int *slot_ref, const x264hip_weight *w );
/* The searches a P request would make WITH a weight, ahead of the request (encoder/slicetype.c:855-867: the first request of a list-0
 * field as a P frame brings the weight x264_weights_analyse arrives at, :284-501 -- a function of the two pictures, which a caller who
 * has their totals and the two cost sums can evaluate as soon as the sums are in): pair i = frame slot_fenc[i] searched on slot_ref[i]
 * weighted by w[i], into a second list-0 field of that distance, and the P cell over it.  An x264hip_frame_cost call that first-triggers
 * the field with the SAME weight takes both over without a launch; any other request ignores them.  The B cells that would read one of
 * these fields -- as the frame's own list-0 field or as the list-1 reference's vectors (slicetype.c:629) -- are evaluated over them as
 * well, into a second spare of the cell.  Never changes results.
 * x264hip_weighted_stats: searches enqueued this way, fields a request took over, cells evaluated over them, B cells of those used. */
int  x264hip_prefetch_weighted_fields( x264hip_ctx *ctx, int n, const int *slot_fenc, const int *slot_ref, const x264hip_weight *w );
int  x264hip_weighted_stats( x264hip_ctx *ctx, uint64_t out[4] );

/* getters (device -> host); sizes in elements: n_mb = mb_w*mb_h */
int  x264hip_get_lowres( x264hip_ctx *ctx, int slot, int plane, void *dst, int dst_stride ); /* incl. 32 px border */
int  x264hip_get_mvs( x264hip_ctx *ctx, int slot, int list, int dist_minus1, int16_t *mvs, int *mv_costs );
int  x264hip_get_lowres_costs( x264hip_ctx *ctx, int slot, int dist_p0, int dist_p1, uint16_t *costs, int *row_satds );
int  x264hip_get_intra_costs( x264hip_ctx *ctx, int slot, uint16_t *intra_costs );
int  x264hip_get_inv_qscale( x264hip_ctx *ctx, int slot, uint16_t *inv_qscale );
int  x264hip_geometry( x264hip_ctx *ctx, int *mb_w, int *mb_h, int *lowres_stride );

/* Speculative batch: enqueue, without waiting, everything that only depends on the pixels of the given
 * frames: for every listed frame b and distance d in [1, bframes+1] the unweighted list-0 search
 * (b -> b-d) and list-1 search (b -> b+d) whose reference is also listed.  Later x264hip_frame_cost
 * calls pick the finished fields up instead of searching.  Never changes results. */
int  x264hip_prefetch( x264hip_ctx *ctx, const int *slots, const int *frame_numbers, int n );

/* What the caller's decisions looked like lately, as a hint for the speculation above: frames anchor_frame + k * period are expected to
 * be coded as P (anchors), the period - 1 frames between two of them as B-frames (period 0 = no expectation).  Which fields and cells
 * the decision flow asks for depends on a frame's position between its anchors; the context keeps, per ( period, position ), how many
 * frames asked for each (list, distance) field class and each (d0, d1) cell class, and -- once enough frames of a position have come
 * and gone -- speculates for a frame at that position only the classes such frames did ask for.  A request for anything that was not
 * speculated is served on demand as always.  Never changes results. */
int  x264hip_gop_hint( x264hip_ctx *ctx, int anchor_frame, int period );

/* The two halves of x264hip_prefetch separately: flags = X264HIP_PREFETCH_CELLS_ONLY skips the searches (the fields are expected
 * to arrive through x264hip_import_field) and only enqueues the speculative cost cells over the fields that exist. */
#define X264HIP_PREFETCH_CELLS_ONLY 1
int  x264hip_prefetch_ex( x264hip_ctx *ctx, const int *slots, const int *frame_numbers, int n, int flags );

/* ---- one lookahead window over several GPUs (x264_amd/shard.py) -----------------------------------------------------
 * The frames of a window are dealt round-robin to the ranks of a node.  The rank that owns frame b runs b's unweighted searches
 * (x264hip_search_fields) AND b's speculative cost cells (x264hip_spec_cells); what a B cell reads from another rank's frame -- the
 * list-1 reference's own list-0 vectors, encoder/slicetype.c:629-642 -- travels as a packed field (x264hip_export_field /
 * x264hip_import_field, n_mb x { mvx | mvy << 16, mv cost } int32 pairs).  Only per-cell SUMMARIES reach the deciding rank
 * (x264hip_export_cells -> x264hip_import_cells): the five sums of slicetype_frame_cost (slicetype.c:946-991) and the row sums,
 * X264HIP_CELL_SUMMARY_INTS( mb_h ) ints per cell.  There the fields count as searched elsewhere (x264hip_fields_remote) and the
 * cells answer x264hip_frame_cost exactly like cells speculated locally; the per-block maps stay with the owner until somebody needs
 * them: MB-tree propagation reads lowres_costs and the vectors of the cells finally chosen -- x264hip_cells_missing tells which of
 * them are not local, x264hip_export_cell_map / x264hip_import_cell_map move one ([3][n_mb] int32: lowres_costs, list-0 vectors,
 * list-1 vectors).  Anything else that needs absent data (a cell evaluated on demand, a getter, a cell MB-tree uses that was not
 * fetched) recomputes it locally: results are identical by construction (a search is a pure function of its two frames).
 * x264hip_field_classes / x264hip_cell_classes: which (list, distance) / (d0, d1) classes the deciding context would speculate, so
 * that it can tell the others (cell class: 0 = none, 1 = without, 2 = with the list-1 reference's vectors, 3 = both ways where that reference's
 * field exists).  x264hip_cells_missing: 0 = the map is here, 1 = with its owner, 2 = the map of the cell's SPARE half is with its owner. */
int  x264hip_search_fields( x264hip_ctx *ctx, int n, const int *slot_b, const int *slot_ref, const int *list, const int *dist_minus1 );
int  x264hip_export_field( x264hip_ctx *ctx, int slot, int list, int dist_minus1, void *dst_dev );
int  x264hip_import_field( x264hip_ctx *ctx, int slot, int list, int dist_minus1, const void *src_dev );
int  x264hip_field_classes( x264hip_ctx *ctx, unsigned *mask_l0, unsigned *mask_l1 );
int  x264hip_cell_classes( x264hip_ctx *ctx, unsigned char *cell_class /* [(bframes+2) * (bframes+2)], index d0 * (bframes+2) + d1 */ );
/* What the caller can say ahead of time about its decision flow: which cell classes (cell_allowed[d0 * (bframes + 2) + d1] != 0;
 * NULL = all) and which field classes (bit d - 1 of mask_l0 / mask_l1 = distance d in list 0 / 1) it can ever ask for.  Classes
 * outside are never speculated (x264hip_prefetch*, x264hip_spec_cells, x264hip_cell_classes, x264hip_field_classes); a request for
 * one is still served, on demand, so a wrong statement costs time and never changes a result.  x264hip_lookahead_open states the
 * classes of its own flow (slicetype.c:1062-1095 with B-pyramid: the middle frame of a run splits it into two halves, so a B-frame
 * never sees two references further apart than half the longest run except as that middle frame). */
int  x264hip_spec_classes( x264hip_ctx *ctx, const unsigned char *cell_allowed, unsigned mask_l0, unsigned mask_l1 );
/* the requests per class so far -- field_req[list * (bframes + 1) + d - 1], cell_req[d0 * (bframes + 2) + d1] -- and the statement in force
 * (any pointer may be NULL): what the tests hold x264hip_lookahead_open's statement against */
int  x264hip_class_requests( x264hip_ctx *ctx, uint32_t *field_req, uint32_t *cell_req, unsigned char *cell_allowed, unsigned *field_allowed );
typedef struct x264hip_cell_ref
{
    int slot_b, slot_p0, slot_p1;   /* frame handles; dist_p0 == dist_p1 == 0: the frame's intra sums */
    int dist_p0, dist_p1;
    int with_ref1_l0;               /* B cells: X264HIP_CELL_* flags below (0 / 1 as before: without / with the list-1 reference's vectors) */
} x264hip_cell_ref;
#define X264HIP_CELL_WITH_L0 1      /* evaluated with the list-1 reference's own list-0 vectors (slicetype.c:629) */
#define X264HIP_CELL_BOTH 2         /* x264hip_spec_cells: evaluate the cell BOTH ways in one pass -- with the vectors into the cell's own place, without them
                                     * into its spare half (the caller's later request decides which one the cell is) */
#define X264HIP_CELL_SPARE 4        /* x264hip_export_cells / _import_cells / _export_cell_map: the entry refers to the spare half (the evaluation WITHOUT the vectors) */
#define X264HIP_CELL_SUMMARY_INTS( mb_h ) ( 8 + 2 * ( mb_h ) ) /* cost_est, cost_est_aq, intra_mbs, intra_cost_est, intra_cost_est_aq, 3 spare, row sums, intra row sums */
int  x264hip_spec_cells( x264hip_ctx *ctx, int n, const x264hip_cell_ref *cells );
int  x264hip_export_cells( x264hip_ctx *ctx, int n, const x264hip_cell_ref *cells, void *dst_dev );
int  x264hip_import_cells( x264hip_ctx *ctx, int n, const x264hip_cell_ref *cells, const void *src_dev );
int  x264hip_fields_remote( x264hip_ctx *ctx, int n, const int *slot, const int *frame_number, const int *list, const int *dist_minus1 );
int  x264hip_cells_missing( x264hip_ctx *ctx, int n, const x264hip_cell_ref *cells, unsigned char *missing );
int  x264hip_export_cell_map( x264hip_ctx *ctx, const x264hip_cell_ref *cell, void *dst_dev );
int  x264hip_import_cell_map( x264hip_ctx *ctx, const x264hip_cell_ref *cell, const void *src_dev );
/* the HIP stream the context enqueues its searches and cells on (hipStream_t): lets a caller order its own device work -- the
 * collectives of x264_amd/shard.py -- against them without a host synchronisation */
int  x264hip_stream_handle( x264hip_ctx *ctx, void **hip_stream );

/* MB-tree: replaces mbtree_propagate_cost / mbtree_propagate_list of x264_mc_functions_t (common/mc.h:333-338,
 * common/mc.c:511-598) and macroblock_tree_finish (encoder/slicetype.c:1029-1049) for a whole list of steps at once.
 * Runs asynchronously on the context's second stream; x264hip_get_qp_offsets waits for it. */
int  x264hip_mbtree( x264hip_ctx *ctx, const x264hip_mbtree_op *ops, int n );
int  x264hip_get_qp_offsets( x264hip_ctx *ctx, int slot, float *qp_offset );        /* f_qp_offset, n_mb floats */
/* x264_picture_t.prop.quant_offsets for a frame already put (x264_adaptive_quant_frame, ratecontrol.c:318-326,396-397): adds one
 * float per macroblock to f_qp_offset / f_qp_offset_aq and recomputes i_inv_qscale_factor = x264_exp2fix8( offset ).  Done through
 * the host (the maps come back, the offsets are added in FP32 like the reference does, the three maps go up again): meant for the
 * occasional region-of-interest picture, not for every frame.  No effect with aq_mode 0. */
int  x264hip_frame_add_quant_offsets( x264hip_ctx *ctx, int slot, const float *quant_offsets );
int  x264hip_get_propagate_cost( x264hip_ctx *ctx, int slot, uint16_t *propagate ); /* i_propagate_cost, n_mb */
/* slicetype_frame_cost_recalculate (encoder/slicetype.c:999-1024; called by x264_rc_analyse_slice, :2002-2003, and by
 * vbv_frame_cost): cost of the evaluated cell (dist_p0, dist_p1) of frame slot_b under its current quantiser offsets --
 * f_qp_offset (after MB-tree), or f_qp_offset_aq when use_aq_offsets (the reference's choice for B frames).  Rewrites the
 * cell's i_row_satds like the reference and returns the frame sum. */
int  x264hip_frame_cost_recalculate( x264hip_ctx *ctx, int slot_b, int dist_p0, int dist_p1, int use_aq_offsets, int *score );

/* ---- vtable-granular primitives, batched -------------------------------------------------------
 * Device counterparts of x264_pixel_function_t.sad/satd (common/pixel.h:78-84, pixel.c:55-80,265-332)
 * over a whole field of blocks: block i of size_idx sits at
 * raster position i of the fenc plane and is compared with the ref plane displaced by the full-pel
 * mv[i].  All seven partition sizes (PIXEL_16x16 = 0, 16x8, 8x16, 8x8, 8x4, 4x8, PIXEL_4x4 = 6); the compared area must be a multiple
 * of 16 samples in both directions.  Planes are device pointers.  Used for parity and for the SAD/SATD GB/s metric. */
int  x264hip_pixel_cmp_batch( x264hip_ctx *ctx, int satd, int size_idx, const void *fenc_plane, const void *ref_plane,
                              int stride, int blocks_w, int blocks_h, const int16_t *mv_dev, int *out_dev );
/* ---- main-encode motion search ----------------------------------------------------------------------
 * x264_me_search_ref (encoder/me.c:182-798: DIA, HEX, UMH, ESA, TESA) + refine_subpel (:865-992) for a batch of independent
 * requests -- what analyse.c hands to x264_me_search_ref for one partition (x264_me_t, common/me.h:33-56, plus the limits of
 * h->mb.mv_limit_fpel / mv_min_spel / mv_max_spel): luma only, one reference per request, no weights.  Planes are device
 * pointers to pixel (0,0); the reference planes are the four half-pel planes x264_frame_filter produces (common/mc.c:704-784),
 * padded far enough for the limits given.  integral_dev: element (0,0) of the 8x8-sum plane with the reference stride,
 * integral_lower elements from there to the 4x4-sum plane (frame.c:240-256); only read by TESA requests.  cost_mv_dev: the centred
 * cost table of the request's qp (h->cost_mv[qp], analyse.c:151-157) on the device.  out[i] = { mv x, mv y (quarter-pel), cost,
 * cost_mv } as x264_me_search_ref leaves them in x264_me_t.  A wave per request (block costs across its lanes). */
/* The integral planes x264_frame_filter keeps for the exhaustive searches (common/mc.c:424-456, :757-783; frame.c:240-256): for the
 * padded luma plane starting at plane_dev (width x height samples, device memory), sum8[y*stride + x] = the sum of the 8x8 box
 * whose top-left sample is (x, y), modulo 2^16, and sum4 the same for 4x4 boxes.  Entries whose box would leave the plane are not
 * written (the reference leaves partial sums there; no search reads them). */
int  x264hip_integral_init( x264hip_ctx *ctx, const void *plane_dev, intptr_t stride, int width, int height, uint16_t *sum8_dev, uint16_t *sum4_dev );
/* What a reconstructed frame goes through before it serves as a reference, for a whole frame: x264_frame_expand_border,
 * x264_frame_filter (hpel planes + integral planes, common/mc.c:704-784) and x264_frame_expand_border_filtered (common/frame.c:
 * 556-623).  luma_dev: the width x height picture (mod-16 size); planes_dev[0..3]: pixel (0,0) of the padded full / H / V / HV
 * planes (padh / padv samples of border on every side, PADH / PADV of the reference = 32); sum8_dev / sum4_dev: element
 * (-padh,-padv)-relative origin, i.e. the first element of integral planes laid out like the padded luma plane (both NULL: no
 * integral planes). */
int  x264hip_frame_filter( x264hip_ctx *ctx, const void *luma_dev, intptr_t luma_stride, int width, int height, void *const planes_dev[4], intptr_t stride,
                           int padh, int padv, uint16_t *sum8_dev, uint16_t *sum4_dev );
#define X264HIP_ME_MVC_MAX 10
typedef struct x264hip_me_request
{
    int i_pixel;              /* PIXEL_16x16 = 0 .. PIXEL_4x4 = 6 */
    int me_method;            /* X264_ME_DIA = 0, HEX, UMH, ESA, TESA = 4 */
    int subpel_refine;        /* h->mb.i_subpel_refine (0..11) */
    int me_range;
    int mbcmp_satd, fpelcmp_satd; /* h->pixf.mbcmp / fpelcmp are SATD (encoder.c:1409-1427) */
    int x, y;                 /* block origin in luma samples */
    int mvp[2];               /* quarter-pel predictor */
    int lim_min[2], lim_max[2];   /* h->mb.mv_limit_fpel */
    int spel_min[2], spel_max[2]; /* h->mb.mv_min_spel / mv_max_spel */
    int n_mvc;                /* candidates in mvc (x264_me_search_ref's mvc / i_mvc) */
    int16_t mvc[X264HIP_ME_MVC_MAX][2];
} x264hip_me_request;
int  x264hip_me_search_batch( x264hip_ctx *ctx, int n, const x264hip_me_request *reqs, const void *fenc_plane_dev, intptr_t fenc_stride,
                              const void *const ref_planes_dev[4], intptr_t ref_stride, const uint16_t *integral_dev, intptr_t integral_lower,
                              const uint16_t *cost_mv_dev, int *out );
/* The same for a request table and a result array that live ON the device (requests generated there): enqueued on the context's stream
 * (x264hip_synchronize to wait), nothing crosses the host link.  Every request is searched with me_method; me_range_max bounds their
 * me_range.  The call above moves 116 bytes per request over PCIe, which is what its rate is bound by. */
int  x264hip_me_search_batch_dev( x264hip_ctx *ctx, int n, const x264hip_me_request *reqs_dev, const void *fenc_plane_dev, intptr_t fenc_stride,
                                  const void *const ref_planes_dev[4], intptr_t ref_stride, const uint16_t *integral_dev, intptr_t integral_lower,
                                  const uint16_t *cost_mv_dev, int me_method, int me_range_max, int *out_dev );

/* The block metrics of x264_pixel_function_t that only the main encode calls, over a raster of blocks_w x blocks_h blocks of size_idx
 * (PIXEL_16x16 = 0 .. PIXEL_4x4 = 6, common/pixel.h:37-59) of device-resident planes sharing one stride; block (x, y) starts at
 * pixel (x*w, y*h).  out_dev[y*blocks_w + x] (device, uint64):
 *   SSD          ssd[size]( a, b )                   common/pixel.c:85-151     all 7 sizes
 *   SA8D         sa8d[size]( a, b )                  :334-381                  16x16, 8x8
 *   VAR          var[size]( a ) = sum | sqr << 32    :183-201                  16x16, 8x16, 8x8
 *   HADAMARD_AC  hadamard_ac[size]( a )              :383-435                  16x16, 16x8, 8x16, 8x8
 *   VSAD         vsad( a, stride, h )                :716-723                  16 wide: sizes 16x16, 16x8 (h rows)
 *   ASD8         asd8( a, b, h )                     :747-754                  8 wide: sizes 8x16, 8x8
 * b_plane is ignored (may be NULL) for the one-plane metrics.  Any other metric / size pair: X264HIP_EINVAL. */
#define X264HIP_METRIC_SSD 0
#define X264HIP_METRIC_SA8D 1
#define X264HIP_METRIC_VAR 2
#define X264HIP_METRIC_HADAMARD_AC 3
#define X264HIP_METRIC_VSAD 4
#define X264HIP_METRIC_ASD8 5
int  x264hip_pixel_metric_batch( x264hip_ctx *ctx, int metric, int size_idx, const void *a_plane, const void *b_plane, intptr_t stride,
                                 int blocks_w, int blocks_h, uint64_t *out_dev );
/* Frame form of sub4x4_dct + quant_4x4 (common/dct.c:157-175, common/quant.c:50-62; SURVEY 8f rank 4, first piece): every
 * 4x4 block of the device-resident plane `fenc` minus the prediction plane `fdec`, transformed and quantised with the host
 * tables mf/bias (16 x udctcoef: uint16 for 8-bit, uint32 for 10-bit).  coefs_dev: [height/4][width/4][16] dctcoef (int16 /
 * int32) in the reference's per-block order, 16-byte aligned; nz_dev: one byte per block = quant_4x4's return value. */
int  x264hip_frame_dct_quant4x4( x264hip_ctx *ctx, const void *fenc, intptr_t fenc_stride, const void *fdec, intptr_t fdec_stride, int width, int height,
                                 const void *mf, const void *bias, void *coefs_dev, uint8_t *nz_dev );
/* Same for sub8x8_dct8 + quant_8x8 (common/dct.c:332-366, common/quant.c:64-72): every 8x8 block; mf/bias hold 64 udctcoef;
 * coefs_dev: [height/8][width/8][64] dctcoef in the reference's per-block order; nz_dev: quant_8x8's return value per block. */
int  x264hip_frame_dct_quant8x8( x264hip_ctx *ctx, const void *fenc, intptr_t fenc_stride, const void *fdec, intptr_t fdec_stride, int width, int height,
                                 const void *mf, const void *bias, void *coefs_dev, uint8_t *nz_dev );
/* x264_mc_functions_t.hpel_filter (common/mc.h:306-307, mc.c:172-196) without the scratch row buffer: the three
 * half-pel planes of `src` (device pointers, element stride).  Like the reference it reads src columns -2..width+2
 * and rows -2..height+2 and also writes dstv columns -2,-1 and width..width+2.  First piece of SURVEY 8(f) rank 3. */
int  x264hip_hpel_filter( x264hip_ctx *ctx, void *dsth, void *dstv, void *dstc, const void *src, intptr_t stride, int width, int height );
/* The three streaming primitives above for up to X264HIP_MULTI_MAX independent planes / plane pairs of ONE geometry in one launch (the
 * arrays hold one device pointer per plane set; strides, sizes and quantiser tables are shared).  The input BASELINE defines for the
 * primitive metric -- the blocks of ONE 4K frame pair -- is a launch of a few microseconds, half of it launch and ramp; the pairs of a
 * lookahead window in one launch run at the rate of one large field.  Same results as n single calls. */
#define X264HIP_MULTI_MAX 16
int  x264hip_pixel_cmp_batch_multi( x264hip_ctx *ctx, int satd, int size_idx, int n_pairs, const void *const *fenc_planes, const void *const *ref_planes, int stride,
                                    int blocks_w, int blocks_h, const int16_t *const *mv_dev, int *const *out_dev );
int  x264hip_hpel_filter_multi( x264hip_ctx *ctx, int n, void *const *dsth, void *const *dstv, void *const *dstc, const void *const *src, intptr_t stride,
                                int width, int height );
int  x264hip_frame_dct_quant4x4_multi( x264hip_ctx *ctx, int n, const void *const *fenc, intptr_t fenc_stride, const void *const *fdec, intptr_t fdec_stride,
                                       int width, int height, const void *mf, const void *bias, void *const *coefs_dev, uint8_t *const *nz_dev );
/* plain device-to-device copy on the context's stream (16-byte aligned): the build's own copy kernel, whose measured
 * GB/s is the second denominator of the SAD/SATD figures next to the vendor peak */
int  x264hip_device_copy( x264hip_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes );
/* x264_mc_functions_t.frame_init_lowres_core (common/mc.h:326-327): device pointers, same arguments */
int  x264hip_frame_init_lowres_core( x264hip_ctx *ctx, const void *src0, void *dst0, void *dsth, void *dstv, void *dstc,
                                     intptr_t src_stride, intptr_t dst_stride, int width, int height );
/* x264_dct_function_t.sub4x4_dct / sub8x8_dct8 (common/dct.h:29-45) and x264_quant_function_t.quant_4x4 /
 * quant_8x8 (common/quant.h:30-34) over n blocks laid out back to back (fenc blocks with stride 16,
 * fdec blocks with stride 32, like FENC_STRIDE/FDEC_STRIDE).  Host pointers; a parity/microbench entry. */
int  x264hip_dct_quant_batch( x264hip_ctx *ctx, int is8x8, int n_blocks, const void *fenc, const void *fdec,
                              const void *mf, const void *bias, void *coefs_out, int *nz_out );

/* The remaining entries of x264_dct_function_t (common/dct.h:29-59) and x264_quant_function_t (common/quant.h:30-45) in batch
 * form: n independent calls on the reference's macroblock-local buffers.  Host pointers.
 *   fenc: n buffers of 16 rows x FENC_STRIDE (16) pixels, fdec: n buffers of 16 rows x FDEC_STRIDE (32) pixels (a kind that works
 *   on a smaller block reads the top-left part); coefs: n x (coefficients of the kind) dctcoef (int16 / int32 for 8 / 10 bit) in
 *   the reference's per-call order.
 *   DCT kinds  0 sub4x4_dct (16)   1 sub8x8_dct (4 x 16)    2 sub16x16_dct (16 x 16)   3 sub8x8_dct8 (64)   4 sub16x16_dct8 (4 x 64)
 *              5 sub8x8_dct_dc (4) 6 sub8x16_dct_dc (8)     7 dct4x4dc (16, in place)  8 dct2x4dc (8 DC values, in place)
 *              (dct.c:47-270, 332-386); the in-place kinds take their input in coefs, fenc / fdec may be NULL.
 *   QUANT kinds 0 quant_4x4  1 quant_8x8  2 quant_4x4x4 (64 coefficients, nz = 4-bit mask)  3 quant_4x4_dc  4 quant_2x2_dc (quant.c:50-104);
 *              mf / bias: 16 (64 for kind 1) udctcoef (uint16 / uint32); the DC kinds use mf_dc / bias_dc.  nz[i] = the entry's return value. */
int  x264hip_dct_batch( x264hip_ctx *ctx, int kind, int n, const void *fenc, const void *fdec, void *coefs );
int  x264hip_quant_batch( x264hip_ctx *ctx, int kind, int n, void *coefs, const void *mf, const void *bias, int mf_dc, int bias_dc, int *nz );
/* x264_pixel_function_t.var2[PIXEL_8x8 / PIXEL_8x16] (common/pixel.c:206-231): the chroma halves of n fenc / fdec buffers as above
 * (U at column 0, V at column stride / 2); var[i] = the return value, ssd[2*i .. 2*i+1] = the two ssd outputs. */
int  x264hip_var2_batch( x264hip_ctx *ctx, int height, int n, const void *fenc, const void *fdec, int *var, int *ssd );
/* x264_pixel_function_t.ads[] (successive elimination, common/pixel.c:756-803): n calls, each over `width` horizontally adjacent
 * candidates; one wave per call keeps the reference's output order (ballot + prefix count).  A call reads sums[sums_off ...] (the
 * integral plane row, x264hip_integral_init), cost_mvx[cost_off ...] and writes the surviving candidate indices to
 * mvs[mvs_off ...]; counts[i] = the return value. */
typedef struct x264hip_ads_call
{
    int n_dc, delta, width, thresh;  /* n_dc 1 / 2 / 4 = ads1 / ads2 / ads4 */
    int enc_dc[4];
    long long sums_off, cost_off, mvs_off;
} x264hip_ads_call;
int  x264hip_ads_batch( x264hip_ctx *ctx, int n, const x264hip_ads_call *calls, const uint16_t *sums, size_t n_sums, const uint16_t *cost_mvx, size_t n_cost,
                        int16_t *mvs, size_t n_mvs, int *counts );

/* ---- vtable-shaped boundary -------------------------------------------------------------------------------------------------
 * The reference fills its function tables once per encoder: x264_pixel_init( cpu, &h->pixf ) (common/pixel.h:146-147),
 * x264_mc_init( cpu, &h->mc, cpu_independent ) (common/mc.h:342-343), x264_dct_init (common/dct.h:72-73), x264_quant_init
 * (common/quant.h:72-73), at encoder/encoder.c:1655-1667.  What this library can stand behind such a table:
 *
 *  - x264_mc_functions_t: the members that work on whole planes or macroblock rows -- plane_copy (mc.h:292), hpel_filter
 *    (:306-307), frame_init_lowres_core (:326-327), mbtree_propagate_cost / mbtree_propagate_list (:333-337).  x264hip_mc_fill
 *    writes functions with EXACTLY the reference's signatures into the struct below (same member names, `pixel` = uint8_t or
 *    uint16_t by the context's bit depth, as in the reference's bit-depth templating, common/common.h:33).  A maintainer calls it
 *    right after x264_mc_init and copies the five pointers into h->mc.  The pointers the encoder passes are host pointers: every
 *    call stages its operands through device memory, so these members are drop-in correct, not fast; the fast forms are the
 *    x264hip_* entries above that take device pointers (x264hip_hpel_filter, x264hip_frame_init_lowres_core, x264hip_mbtree, ...).
 *    The functions have no context argument (neither have the reference's), so the library keeps a process-wide registry: one bound
 *    context per bit depth (the last x264hip_*_fill call of that depth), plus contexts registered for an encoder handle
 *    (x264hip_mc_bind_handle: mbtree_propagate_list receives the x264_t * and reads the picture geometry of the context registered for
 *    it).  The members may be called from any thread: calls are serialised by the registry lock and a bound context cannot be closed
 *    under a running call.  They return like the originals; a device failure latches the context like x264_opencl_t.b_fatal_error.
 *  - x264_pixel_function_t / x264_dct_function_t / x264_quant_function_t: every member is a per-block call (an 8x8 SAD reads 128
 *    bytes and returns an int); behind a host function pointer each call costs a PCIe round trip (~50 us) for ~10 ns of work, and
 *    the reference calls them millions of times per frame from a serial loop.  The FAST forms are therefore the batch entries --
 *    x264hip_pixel_cmp_batch (sad / satd, all 7 sizes), x264hip_pixel_metric_batch (ssd, sa8d, var, hadamard_ac, vsad, asd8),
 *    x264hip_var2_batch, x264hip_ads_batch, x264hip_dct_batch (all 9 dctf entries), x264hip_quant_batch (all 5 quantf entries) --
 *    and the hot callers of those tables on this path (slicetype_mb_cost, x264_me_search_ref) run on the device as a whole behind
 *    the coarse hook (x264hip_frame_cost), which is where the reference's own accelerator boundary is (slicetype.c:878-897).
 *    For completeness of the table-shaped boundary x264hip_pixel_fill / x264hip_dct_fill / x264hip_quant_fill hand out members with
 *    the reference's exact signatures as well (same member names; `pixel` / `dctcoef` / `udctcoef` by the context's bit depth): each
 *    call stages its one block through the batch entry -- drop-in correct, three orders of magnitude slower than the C version. */
typedef struct x264hip_mc_functions
{
    void (*plane_copy)( void *dst, intptr_t i_dst, void *src, intptr_t i_src, int w, int h );
    void (*hpel_filter)( void *dsth, void *dstv, void *dstc, void *src, intptr_t i_stride, int i_width, int i_height, int16_t *buf );
    void (*frame_init_lowres_core)( void *src0, void *dst0, void *dsth, void *dstv, void *dstc, intptr_t src_stride, intptr_t dst_stride, int width, int height );
    void (*mbtree_propagate_cost)( int16_t *dst, uint16_t *propagate_in, uint16_t *intra_costs, uint16_t *inter_costs, uint16_t *inv_qscales, float *fps_factor, int len );
    void (*mbtree_propagate_list)( void *h, uint16_t *ref_costs, int16_t (*mvs)[2], int16_t *propagate_amount, uint16_t *lowres_costs, int bipred_weight, int mb_y,
                                   int len, int list );
} x264hip_mc_functions;
int  x264hip_mc_fill( x264hip_ctx *ctx, x264hip_mc_functions *pf );
/* x264_dct_function_t (common/dct.h:29-59): the forward transforms */
typedef struct x264hip_dct_functions
{
    void (*sub4x4_dct)( void *dct /* dctcoef[16] */, void *pix1, void *pix2 );
    void (*sub8x8_dct)( void *dct /* dctcoef[4][16] */, void *pix1, void *pix2 );
    void (*sub8x8_dct_dc)( void *dct /* dctcoef[4] */, void *pix1, void *pix2 );
    void (*sub8x16_dct_dc)( void *dct /* dctcoef[8] */, void *pix1, void *pix2 );
    void (*sub16x16_dct)( void *dct /* dctcoef[16][16] */, void *pix1, void *pix2 );
    void (*sub8x8_dct8)( void *dct /* dctcoef[64] */, void *pix1, void *pix2 );
    void (*sub16x16_dct8)( void *dct /* dctcoef[4][64] */, void *pix1, void *pix2 );
    void (*dct4x4dc)( void *d /* dctcoef[16] */ );
    void (*dct2x4dc)( void *dct /* dctcoef[8] */, void *dct4x4 /* dctcoef[8][16] */ );
} x264hip_dct_functions;
int  x264hip_dct_fill( x264hip_ctx *ctx, x264hip_dct_functions *pf );
/* x264_quant_function_t (common/quant.h:30-45): the quantisers */
typedef struct x264hip_quant_functions
{
    int (*quant_8x8)( void *dct /* dctcoef[64] */, void *mf /* udctcoef[64] */, void *bias );
    int (*quant_4x4)( void *dct, void *mf, void *bias );
    int (*quant_4x4x4)( void *dct /* dctcoef[4][16] */, void *mf, void *bias );
    int (*quant_4x4_dc)( void *dct, int mf, int bias );
    int (*quant_2x2_dc)( void *dct /* dctcoef[4] */, int mf, int bias );
} x264hip_quant_functions;
int  x264hip_quant_fill( x264hip_ctx *ctx, x264hip_quant_functions *pf );
/* x264_pixel_function_t (common/pixel.h:78-146): sad / ssd / satd over the sizes PIXEL_16x16 .. PIXEL_4x4, sa8d[PIXEL_16x16], sa8d[PIXEL_8x8],
 * var[PIXEL_16x16 / PIXEL_8x16 / PIXEL_8x8], hadamard_ac[PIXEL_16x16 .. PIXEL_8x8], the x3 / x4 forms, vsad, asd8, var2, ads and the
 * mbcmp / fpelcmp aliases; entries the reference leaves empty are NULL */
typedef struct x264hip_pixel_functions
{
    int (*sad[8])( void *pix1, intptr_t i_stride1, void *pix2, intptr_t i_stride2 );
    int (*ssd[8])( void *pix1, intptr_t i_stride1, void *pix2, intptr_t i_stride2 );
    int (*satd[8])( void *pix1, intptr_t i_stride1, void *pix2, intptr_t i_stride2 );
    int (*sa8d[4])( void *pix1, intptr_t i_stride1, void *pix2, intptr_t i_stride2 );
    uint64_t (*var[4])( void *pix, intptr_t stride );
    uint64_t (*hadamard_ac[4])( void *pix, intptr_t stride );
    /* several candidates against one source block (common/pixel.h:107-110, pixel.c:441-516): fenc has FENC_STRIDE (16), the candidates share i_stride */
    void (*sad_x3[7])( void *fenc, void *pix0, void *pix1, void *pix2, intptr_t i_stride, int scores[3] );
    void (*sad_x4[7])( void *fenc, void *pix0, void *pix1, void *pix2, void *pix3, intptr_t i_stride, int scores[4] );
    void (*satd_x3[7])( void *fenc, void *pix0, void *pix1, void *pix2, intptr_t i_stride, int scores[3] );
    void (*satd_x4[7])( void *fenc, void *pix0, void *pix1, void *pix2, void *pix3, intptr_t i_stride, int scores[4] );
    int (*vsad)( void *pix, intptr_t stride, int height );                                         /* pixel.c:716-723, height 8 or 16 */
    int (*asd8)( void *pix1, intptr_t stride1, void *pix2, intptr_t stride2, int height );           /* pixel.c:747-754, height 8 or 16 */
    int (*var2[4])( void *fenc, void *fdec, int ssd[2] );                                          /* [PIXEL_8x16], [PIXEL_8x8] (pixel.c:206-231) */
    int (*ads[7])( int enc_dc[4], uint16_t *sums, int delta, uint16_t *cost_mvx, int16_t *mvs, int width, int thresh ); /* [PIXEL_16x16] ads4, [PIXEL_16x8] ads2, [PIXEL_8x8] ads1 */
    /* what mbcmp_init (encoder/encoder.c:1409-1427) copies: satd or sad by the context's subme (> 1) and, for the full-pel tables, --me tesa */
    int (*mbcmp[8])( void *pix1, intptr_t i_stride1, void *pix2, intptr_t i_stride2 );
    int (*mbcmp_unaligned[8])( void *pix1, intptr_t i_stride1, void *pix2, intptr_t i_stride2 );
    int (*fpelcmp[8])( void *pix1, intptr_t i_stride1, void *pix2, intptr_t i_stride2 );
    void (*fpelcmp_x3[7])( void *fenc, void *pix0, void *pix1, void *pix2, intptr_t i_stride, int scores[3] );
    void (*fpelcmp_x4[7])( void *fenc, void *pix0, void *pix1, void *pix2, void *pix3, intptr_t i_stride, int scores[4] );
    int (*sad_aligned[8])( void *pix1, intptr_t i_stride1, void *pix2, intptr_t i_stride2 );
    /* NOT handed out, deliberately: intra_*_x3_* / intra_*_x9_* (they write their predictions into the encoder's fdec buffer: intra analysis
     * of the main encode, SURVEY 2 out of scope; the lookahead's own intra costs are intra_kernel behind x264hip_frame_put), ssim[] /
     * ssim_4x4x2_core / ssim_end4 / ssd_nv12_core (quality statistics, not on the hot path), sa8d_satd (assembly-only pairing). */
} x264hip_pixel_functions;
int  x264hip_pixel_fill( x264hip_ctx *ctx, x264hip_pixel_functions *pf );
/* registers the context for an encoder handle: mbtree_propagate_list( h, ... ) then works on THAT context's picture geometry */
int  x264hip_mc_bind_handle( x264hip_ctx *ctx, const void *encoder_handle );
void x264hip_mc_unbind( x264hip_ctx *ctx ); /* every binding of the context; done by x264hip_close as well */

/* Test entry for x264_mc_functions_t.mc_luma / get_ref (common/mc.c:198-249) on the lowres planes: out[i] = the 8x8 block (64 pixels of
 * the context's bit depth, row-major) of frame `slot` at lowres position (x, y) displaced by (mvx, mvy) quarter-pels, weighted by *w if
 * w != NULL -- produced by the tap arithmetic every candidate of the search and cell kernels goes through (strip copy of the four
 * half-pel planes).  tools/checkasm.c:1226-1290 sweeps the reference's mc_luma the same way; tests/test_gpu_parity.py does it here over
 * all 16 phases, the picture's corners and the furthest displacements the padding allows.  EINVAL for a block that leaves the padded
 * planes.  Host pointers; synchronous. */
typedef struct x264hip_mc_probe
{
    int x, y, mvx, mvy;
} x264hip_mc_probe;
int  x264hip_mc_luma_probe( x264hip_ctx *ctx, int slot, int n, const x264hip_mc_probe *req, const x264hip_weight *w, void *out );

/* timing of the most recent search launch in ms (HIP events on the context's stream) and counters */
int  x264hip_last_search_ms( x264hip_ctx *ctx, float *ms, int *n_searches, int *n_blocks );
int  x264hip_counters( x264hip_ctx *ctx, uint64_t *out, int n );
/* Per-launch HIP-event timing of the search kernel on the context's stream.  Returns the totals gathered
 * since profiling was last (re)enabled; enable = 1/0 switches it and clears the totals, -1 only reads. */
int  x264hip_search_profile( x264hip_ctx *ctx, int enable, double *total_ms, uint64_t *launches, uint64_t *searches ); /* [0] searches [1] cells [2] cache hits [3] frames */
/* ... and the launches among them that cannot fill the chip (fewer waves than the device has wave slots: as long as their dependency chain
 * whatever runs them), for themselves; the totals above then hold the chip-filling launches of the throughput kernel only.  Read before x264hip_search_profile, which resets both. */
int  x264hip_search_profile_latency( x264hip_ctx *ctx, double *total_ms, uint64_t *launches, uint64_t *searches );
/* The same window's totals for the cost cell launches (cell kernels + their sums): together with the searches, the device work a
 * window shard spreads over ranks (bench.py: window_shard.amdahl). */
int  x264hip_cell_profile( x264hip_ctx *ctx, double *total_ms, uint64_t *launches, uint64_t *cells );
/* x264hip_search_profile( enable | 2 ) also puts an event pair around every kernel of the ingest and cost-cell launches; per class:
 * summed device time, launches, and the frames (ingest classes) / cells (cell classes) they worked on -- what bench.py prices the
 * kernels beside the search against their rooflines with.  Arrays of X264HIP_KPROF_CLASSES entries. */
#define X264HIP_KPROF_LOWRES 0      /* lowres_tiles_kernel: planes + strip copy (frame_init_lowres_core + border) */
#define X264HIP_KPROF_AQ 1          /* aq_kernel (adaptive_quant_frame: var_16x16 / 8x8 per macroblock) */
#define X264HIP_KPROF_INTRA 2       /* intra_kernel: the ten intra modes of an 8x8 lowres block */
#define X264HIP_KPROF_CELL_P 3      /* cell_p_kernel */
#define X264HIP_KPROF_CELL_B 4      /* cell_b_kernel: the three bidirectional candidates of every block */
#define X264HIP_KPROF_CELL_REDUCE 5 /* cell_reduce_kernel */
#define X264HIP_KPROF_CLASSES 6
int  x264hip_kernel_profile( x264hip_ctx *ctx, double *total_ms, uint64_t *launches, uint64_t *units );


/* ==================================================================================================
 * Host-side lookahead: the reference's own control flow (stays on the CPU) driving the
 * device evaluations above.  Mirrors, with the same pacing as x264_encoder_encode
 * (encoder/encoder.c:3417-3454):
 *   x264hip_lookahead_put_frame  <- x264_frame_init_lowres + x264_lookahead_put_frame (encoder/lookahead.c:192)
 *   x264hip_lookahead_get_frame  <- x264_lookahead_get_frames (lookahead.c:223) + x264_frame_shift of
 *                                   h->frames.current, i.e. x264_slicetype_decide / x264_slicetype_analyse
 *                                   (encoder/slicetype.c:1745,1473) with slicetype_frame_cost (:836) memoisation,
 *                                   first-trigger weights (x264_weights_analyse :284, lookahead mode), scenecut,
 *                                   B-adapt fast/trellis paths, and the MB-tree / final-cost evaluation order.
 * Decisions (slice types) and every i_cost_est cell are identical to the reference's with threads=1.
 * ================================================================================================== */
typedef struct x264hip_lookahead x264hip_lookahead;

typedef struct x264hip_la_params
{
    x264hip_params dev;       /* dev.max_frames is derived when 0 */
    int keyint_max, keyint_min;
    int scenecut_threshold;
    int b_adapt;              /* 0 none, 1 fast, 2 trellis */
    int b_pyramid;            /* 0 none, 1 strict, 2 normal */
    int rc_lookahead;
    int mb_tree;              /* drives the evaluation order, do_edges and key-frame analysis; propagation itself is not computed */
    int weightp;              /* param.analyse.i_weighted_pred (0..2) */
    int open_gop;
    int frame_refs;           /* param.i_frame_reference (B-pyramid compatibility rule, slicetype.c:1819) */
    int psy;                  /* param.analyse.b_psy (slicetype.c:1512) */
    int rc_is_cqp;            /* rc.i_rc_method == X264_RC_CQP: skips the final cost evaluation (slicetype.c:1899) */
    int fps_num, fps_den;     /* constant frame rate (f_duration of every frame, slicetype.c:1767-1771); 0 -> 25/1 */
    float qcompress;          /* param.rc.f_qcompress (MB-tree strength, slicetype.c:1038); 0 -> 0.6 */
    int vbv;                  /* param.rc.i_vbv_buffer_size != 0 after validation: with rc_lookahead > 0 turns on the VBV lookahead
                               * (vbv_lookahead, slicetype.c:1224-1286; key-frame analysis, lookahead.c:140-141), the B-frame and
                               * intra evaluations slicetype_decide adds for the row sums (:1916-1934), an MB-tree finish for every
                               * reference (:1087-1088) and the lookahead delay of encoder.c:1607-1608 */
    int vfr_input;            /* param.b_vfr_input: frame durations come from the time stamps given to x264hip_lookahead_put_frame_pts
                               * (slicetype.c:1755-1771; MB-tree weighs frames by duration, :1031,1063,1098-1101), and the delay grows
                               * by one frame (encoder.c:1612) */
    int timebase_num, timebase_den; /* param.i_timebase_num/den for vfr_input; 0 -> fps_den / fps_num (encoder.c:1119-1123) */
    int intra_refresh;        /* param.b_intra_refresh: no key frames after the first (slicetype.c:1405,1506,1681,1831); the column
                               * bookkeeping and the row-sum correction of x264_rc_analyse_slice (:2015-2032) stay in the encoder */
} x264hip_la_params;



/* pluggable evaluation backend (same contracts as the x264hip_* device entry points) */
typedef struct x264hip_backend
{
    void *user;
    int (*frame_put)( void *user, int slot, const void *luma, int stride, int is_device );
    int (*frame_stats)( void *user, int slot, uint64_t *pixel_sum, uint64_t *pixel_ssd );
    int (*weight_cost)( void *user, int slot_fenc, int slot_ref, const x264hip_weight *w, unsigned *cost );
    int (*frame_cost)( void *user, int slot_p0, int slot_p1, int slot_b, int dist_p0, int dist_p1, const int do_search[2],
                       const x264hip_weight *w, int with_intra, int ref1_l0_valid, x264hip_cost *out );
    int (*prefetch)( void *user, const int *slots, const int *frame_numbers, int n ); /* may be NULL */
    int (*mbtree)( void *user, const x264hip_mbtree_op *ops, int n );                   /* may be NULL: no propagation */
    int (*get_qp_offsets)( void *user, int slot, float *qp_offset );                    /* may be NULL */
    int (*frame_put_batch)( void *user, int n, const int *slots, const void *const *luma_dev, int stride ); /* may be NULL */
    int (*prefetch_weight_costs)( void *user, int n, const int *slot_fenc, const int *slot_ref, const x264hip_weight *w ); /* may be NULL */
    /* the two below are needed with vbv only (may be NULL otherwise); contracts of x264hip_frame_cost_recalculate and of the
     * row_satds output of x264hip_get_lowres_costs */
    int (*frame_cost_recalculate)( void *user, int slot_b, int dist_p0, int dist_p1, int use_aq_offsets, int *score );
    int (*get_row_satds)( void *user, int slot, int dist_p0, int dist_p1, int *row_satds );
    /* frame_put with the 4:2:0 chroma planes (contract of x264hip_frame_put's cb / cr / cstride): needed by
     * x264hip_lookahead_put_picture, may be NULL otherwise */
    int (*frame_put_yuv)( void *user, int slot, const void *luma, int stride, const void *cb, const void *cr, int cstride, int is_device );
    /* adds the caller's per-macroblock offsets to the AQ maps of a frame just put (f_qp_offset, f_qp_offset_aq, i_inv_qscale_factor):
     * contract of x264hip_frame_add_quant_offsets; needed only for pictures that carry quant_offsets */
    int (*add_quant_offsets)( void *user, int slot, const float *quant_offsets );
    int (*frame_put_batch_yuv)( void *user, int n, const int *slots, const void *const *luma_dev, int stride, const void *const *cb_dev,
                                const void *const *cr_dev, int cstride ); /* may be NULL: the pictures go in one by one */
    int (*gop_hint)( void *user, int anchor_frame, int period ); /* contract of x264hip_gop_hint, called in front of prefetch; may be NULL */
    /* contract of x264hip_flush: called when the last delayed frame of a flush has been handed out, so that no work the lookahead
     * asked for stays queued in the backend behind the end of the stream; may be NULL */
    int (*flush)( void *user );
    /* contract of x264hip_prefetch_weighted_fields: the list-0 searches (and P cells) of the pairs whose weight the lookahead has found it
     * will keep, ahead of the requests; may be NULL */
    int (*prefetch_weighted_fields)( void *user, int n, const int *slot_fenc, const int *slot_ref, const x264hip_weight *w );
} x264hip_backend;

typedef struct x264hip_la_frame
{
    int frame;                /* display-order index (i_frame) */
    int type;                 /* X264_TYPE_*: 1 IDR, 2 I, 3 P, 4 BREF, 5 B */
    int bframes;              /* i_bframes of a non-B frame */
    int keyframe;
    int cost_est[X264HIP_BFRAME_MAX + 2][X264HIP_BFRAME_MAX + 2];    /* fenc->i_cost_est, -1 = never evaluated */
    int cost_est_aq[X264HIP_BFRAME_MAX + 2][X264HIP_BFRAME_MAX + 2];
    int intra_mbs[X264HIP_BFRAME_MAX + 2];
} x264hip_la_frame;

/* Opens a device context (x264hip_open) and the host logic on top of it.  No CPU fallback. */
int  x264hip_lookahead_open( x264hip_lookahead **out, int device, const x264hip_la_params *params );
/* The device lookahead with its speculative submissions routed through the caller: hook( user, slots, frame_numbers, n ) is called
 * where the host logic would call x264hip_prefetch (same arguments: every resident frame the next decisions can reach) and is
 * expected to make those fields and cells available by its own means -- x264_amd/shard.py searches them on several GPUs and
 * finishes with x264hip_import_field + x264hip_prefetch_ex( CELLS_ONLY ).  Never changes results. */
typedef int (*x264hip_prefetch_hook)( void *user, const int *slots, const int *frame_numbers, int n );
int  x264hip_lookahead_open_hooked( x264hip_lookahead **out, int device, const x264hip_la_params *params, x264hip_prefetch_hook hook, void *user );
/* Called with the step list right before the host logic hands it to x264hip_mbtree: the window shard fetches the maps of the cells
 * the propagation is about to read (x264hip_cells_missing / x264hip_import_cell_map).  NULL = none. */
typedef int (*x264hip_mbtree_hook)( void *user, const x264hip_mbtree_op *ops, int n );
int  x264hip_lookahead_set_mbtree_hook( x264hip_lookahead *la, x264hip_mbtree_hook hook, void *user );
/* How many frames beyond the reach of the next decision one speculative submission covers when more than that are queued (batch
 * ingest); 0 = the default (256; 64 for a hooked lookahead, whose chunks are rounds of collectives).  Never changes results. */
int  x264hip_lookahead_set_chunk( x264hip_lookahead *la, int frames );
/* Same host logic over a caller-supplied backend (plugin / test hook). */
int  x264hip_lookahead_open_backend( x264hip_lookahead **out, const x264hip_la_params *params, const x264hip_backend *backend );
void x264hip_lookahead_close( x264hip_lookahead *la );
int  x264hip_lookahead_reset( x264hip_lookahead *la ); /* start a new sequence on the same context */
/* pictures put but not yet returned by get_frame (the lookahead's share of x264_encoder_delayed_frames, encoder.c:4480-4500) */
int  x264hip_lookahead_delayed_frames( x264hip_lookahead *la );
x264hip_ctx *x264hip_lookahead_ctx( x264hip_lookahead *la ); /* NULL for plugin backends */
/* The cell classes ( cell_allowed[d0 * (bframes + 2) + d1] ) and field classes ( bit d - 1 of mask_l0 / mask_l1 ) the decisions of a
 * lookahead with these parameters can ever ask for -- what x264hip_lookahead_open states to its context (x264hip_spec_classes).  Pure
 * host arithmetic, no device: with B-pyramid a run of B-frames is always split at its middle frame (slicetype.c:1062-1095,
 * :1120-1160, :1922-1933), which rules out most of the triangle for long runs. */
int  x264hip_lookahead_classes( const x264hip_la_params *params, unsigned char *cell_allowed, unsigned *mask_l0, unsigned *mask_l1 );
int  x264hip_lookahead_delay( x264hip_lookahead *la );       /* h->frames.i_delay */
/* forced_type: X264_TYPE_AUTO (0) normally */
int  x264hip_lookahead_put_frame( x264hip_lookahead *la, const void *luma, int stride, int is_device, int forced_type );
/* n frames at once (display order, all X264_TYPE_AUTO): batched ingest when the backend supports it.  Device pointers, or -- with the
 * HIP backend -- host pointers (x264hip_frame_put_batch: pinned buffers are read where they are, pageable ones staged) */
int  x264hip_lookahead_put_frames( x264hip_lookahead *la, int n, const void *const *luma_dev, int stride );
/* a whole clip of device-resident frames in one call: n frames put and every decided frame taken (out[n], coded order; *n_out of them),
 * paced != 0: one put and one get per frame like x264_encoder_encode (encoder/encoder.c:3300-3440), then the flush; else all frames
 * put first.  Equivalent to the put_frame(s) / get_frame calls it makes. */
int  x264hip_lookahead_run_frames( x264hip_lookahead *la, int n, const void *const *luma_dev, int stride, int paced, x264hip_la_frame *out,
                                   int *n_out );
/* batch form of x264hip_lookahead_put_picture for device-resident 4:2:0 pictures: cb_dev / cr_dev NULL = luma only; types (forced
 * picture types) and pts may be NULL (AUTO / the frame numbers) */
int  x264hip_lookahead_put_pictures( x264hip_lookahead *la, int n, const void *const *luma_dev, int stride, const void *const *cb_dev,
                                     const void *const *cr_dev, int cstride, const int *types, const int64_t *pts );
/* One call = the lookahead part of one x264_encoder_encode call.  flush != 0 once the input has ended.
 * *got = 1 and *out filled when a frame leaves the lookahead (coded order), 0 while the delay fills or at the end. */
/* The whole 4:2:0 picture: adaptive quantisation measures the AC energy of luma AND chroma (ac_energy_mb, ratecontrol.c:258-276), so
 * the chroma planes are needed for the reference's i_inv_qscale_factor / f_qp_offset on real content (luma-only input is treated as
 * flat chroma: no chroma energy).  planes = { Y, Cb, Cr }, strides in samples, host or device pointers (is_device). */
int  x264hip_lookahead_put_picture( x264hip_lookahead *la, const void *const planes[3], const int strides[3], int is_device, int forced_type, int64_t pts );
/* The general form, shaped like the x264_picture_t fields the lookahead reads (x264.h): planes / strides as above (Cb, Cr may be
 * NULL), i_type, i_pts, and prop.quant_offsets -- one float per macroblock added to the adaptive-quantisation offset of the
 * picture (ratecontrol.c:318-326,396-397; NULL = none; host memory, consumed before the call returns). */
typedef struct x264hip_picture
{
    const void *planes[3];
    int strides[3];
    int is_device;
    int i_type;
    int64_t i_pts;
    const float *quant_offsets;
} x264hip_picture;
int  x264hip_lookahead_put( x264hip_lookahead *la, const x264hip_picture *pic );
/* same with the picture's time stamp (x264_picture_t.i_pts, in timebase units); x264hip_lookahead_put_frame uses the frame number */
int  x264hip_lookahead_put_frame_pts( x264hip_lookahead *la, const void *luma, int stride, int is_device, int forced_type, int64_t pts );
int  x264hip_lookahead_get_frame( x264hip_lookahead *la, int flush, x264hip_la_frame *out, int *got );
/* same, additionally copying the frame's f_qp_offset map (mb_w*mb_h floats: the AQ offsets, replaced by the MB-tree output
 * when mb_tree is on; read by rate control, encoder/ratecontrol.c:1761) when qp_offset != NULL and aq_mode != 0 */
int  x264hip_lookahead_get_frame_ex( x264hip_lookahead *la, int flush, x264hip_la_frame *out, int *got, float *qp_offset );

/* What VBV rate control reads from a frame leaving the lookahead besides the above (encoder/ratecontrol.c:1545-1613,
 * 2290-2320, and x264_rc_analyse_slice, slicetype.c:1976-2030):
 *  - i_planned_type / i_planned_satd: types and costs of the following frames in coded order, ended by type 0 (AUTO); filled by
 *    the last analysis that saw the frame as the next non-B frame (or as the key frame just decided); n_planned = 0 for B frames
 *  - the (dist_p0, dist_p1) cell the frame is coded with, satd = the return value of x264_rc_analyse_slice for it, and that cell's
 *    i_row_satds plus the intra row sums i_row_satds[0][0] as x264_rc_analyse_slice leaves them (with MB-tree: rewritten by
 *    slicetype_frame_cost_recalculate); row_satds / row_satds_intra: caller buffers of mb_h ints, may be NULL.  Usable without
 *    VBV too (n_planned = 0 then): satd is what ABR / CRF rate control reads as the frame's complexity.
 * The f_planned_cpb_duration bookkeeping (calculate_durations, slicetype.c:1200-1222) is left to the caller: it is the frame
 * duration for progressive constant-frame-rate input. */
#define X264HIP_LOOKAHEAD_MAX 250 /* X264_LOOKAHEAD_MAX, common/common.h */
typedef struct x264hip_la_vbv
{
    int n_planned;
    int planned_type[X264HIP_LOOKAHEAD_MAX + 1];
    int planned_satd[X264HIP_LOOKAHEAD_MAX + 1];
    int dist_p0, dist_p1;
    int satd;                 /* what x264_rc_analyse_slice returns for the frame (slicetype.c:1976-2009, no intra refresh): the cell's
                               * cost, recalculated under the final f_qp_offset with MB-tree, the AQ-weighted cost otherwise; -1 when the
                               * cell was never evaluated (constant QP) */
} x264hip_la_vbv;
int  x264hip_lookahead_get_frame_vbv( x264hip_lookahead *la, int flush, x264hip_la_frame *out, int *got, float *qp_offset,
                                      x264hip_la_vbv *vbv, int *row_satds, int *row_satds_intra );
/* statistics: [0] slicetype_frame_cost calls, [1] real evaluations, [2] weights analysed, [3] weights kept,
 * wall time in ns spent in [4] backend frame_cost, [5] weights_analyse, [6] backend prefetch + mbtree, [7] the put/get calls in total */
int  x264hip_lookahead_stats( x264hip_lookahead *la, uint64_t *out, int n );

/* ---- one lookahead window over the GPUs of a node, from C (x264_amd/csrc/shard_host.cpp; SURVEY 8e, BASELINE configs[3]) ----------------
 * The host side of the entries above, inside the library: what encoder/slicetype-cl.h:29-42 is to one OpenCL device, for N GPUs.  One
 * process (or thread) per GPU opens a shard with the same parameters and a transport of `world` ranks.  Rank 0 then drives the ordinary
 * lookahead calls on x264hip_shard_lookahead() -- pictures enter through x264hip_shard_put_frames, which also remembers where they
 * are so that they can be broadcast -- while every other rank sits in x264hip_shard_serve() until rank 0 closes its shard.  Frame b's
 * searches and cost cells run on rank b % world; only cell summaries (and, before an MB-tree call, the per-block maps that call reads)
 * reach rank 0, which decides.  Slice types, cost cells and f_qp_offset are those of the single stream.
 * Errors: after every command a status word is max-reduced over the ranks; a rank whose device calls fail keeps issuing the command's
 * collectives (the plan tells every rank the counts) and reports through that word, so the NEXT call fails on every rank -- with its own
 * error where it failed, X264HIP_EPEER elsewhere -- instead of leaving the others inside a collective.
 * The transport: four operations on DEVICE buffers, each enqueued on the given HIP stream (the context's own: searches, exports,
 * collectives and imports are ordered on the device, no host thread waits for a chunk).  x264hip_shard_transport_rccl fills one in over
 * RCCL (xGMI inside a node); librccl.so is opened at run time, X264HIP_ENODEV if it is absent.  A one-rank transport with .loopback set
 * runs every exchange step with the rank as its own peer (tests: the whole path on one GPU; x264hip_shard_loopback_verify).
 * Every exchange buffer is allocated by x264hip_shard_open (a failed open leaves the transport untouched: it stays the caller's to
 * destroy); larger exchanges travel in pieces every rank sizes alike (X264HIP_SHARD_PIECE_BYTES overrides the per-buffer budget). */
#define X264HIP_EPEER    -7   /* window shard: another rank failed (its own error is reported there) */
typedef struct x264hip_shard x264hip_shard;
typedef struct x264hip_shard_transport
{
    void *user;
    int rank, world;
    int loopback;             /* world == 1 only: run the exchange anyway, this rank being its own peer */
    /* buf: `bytes` bytes, from rank `root` to every rank, in place */
    int (*broadcast)( void *user, void *buf, size_t bytes, int root, void *hip_stream );
    /* sbuf: send_bytes[0] bytes for rank 0, then send_bytes[1] for rank 1, ...; rbuf receives recv_bytes[r] bytes from rank r in rank order */
    int (*send_recv)( void *user, const void *sbuf, const size_t *send_bytes, void *rbuf, const size_t *recv_bytes, void *hip_stream );
    /* every rank's `bytes` bytes to rank `root`: rbuf (root only) = world x bytes in rank order */
    int (*gather)( void *user, const void *sbuf, void *rbuf, size_t bytes, int root, void *hip_stream );
    int (*allreduce_max_i32)( void *user, int *buf, int n, void *hip_stream );
    void (*destroy)( void *user ); /* may be NULL; called by x264hip_shard_close */
} x264hip_shard_transport;
int  x264hip_rccl_unique_id( void *id128 /* 128 bytes: ncclUniqueId, from rank 0 to the others by the caller's own means */ );
int  x264hip_shard_transport_rccl( x264hip_shard_transport *t, const void *nccl_unique_id, int rank, int world, int device );
int  x264hip_shard_open( x264hip_shard **out, int device, const x264hip_la_params *params, const x264hip_shard_transport *transport );
x264hip_lookahead *x264hip_shard_lookahead( x264hip_shard *s ); /* rank 0: put / get as usual (pictures through the call below) */
x264hip_ctx *x264hip_shard_ctx( x264hip_shard *s );
/* rank 0: n device-resident pictures, frame numbers first_number .. first_number + n - 1 (display order, counted from 0 like the
 * lookahead counts them), put into the lookahead; the pointers must stay valid until those frames have been returned by get_frame */
int  x264hip_shard_put_frames( x264hip_shard *s, int first_number, int n, const void *const *luma_dev, int stride );
int  x264hip_shard_serve( x264hip_shard *s );                   /* ranks 1 .. world-1: returns when rank 0 closes; X264HIP_OK or the first error */
/* rank 0: start a new sequence (frame numbers from 0 again) on EVERY rank.  Use this instead of x264hip_lookahead_reset on
 * x264hip_shard_lookahead(): the ranks key the pictures, fields and cells they hold by frame number. */
int  x264hip_shard_reset( x264hip_shard *s );
#define X264HIP_SHARD_CHUNKS 0
#define X264HIP_SHARD_FIELDS_SEARCHED 1
#define X264HIP_SHARD_CELLS_EVALUATED 2
#define X264HIP_SHARD_L0_FIELDS 3      /* list-0 fields received from their owners */
#define X264HIP_SHARD_CELLS_IMPORTED 4
#define X264HIP_SHARD_MAPS_FETCHED 5
#define X264HIP_SHARD_FETCH_COMMANDS 6
#define X264HIP_SHARD_BYTES_INPUT 7    /* bytes this rank received / rank 0 sent per peer: pictures, fields, summaries, maps */
#define X264HIP_SHARD_BYTES_L0 8
#define X264HIP_SHARD_BYTES_SUMMARIES 9
#define X264HIP_SHARD_BYTES_MAPS 10
#define X264HIP_SHARD_MAPS_FETCHED_SPARE 11 /* of the maps fetched: the spare half of a cell evaluated both ways (the variant without the list-1 reference's vectors) */
#define X264HIP_SHARD_STATS 12
int  x264hip_shard_status( x264hip_shard *s );                   /* waits for everything enqueued; X264HIP_OK, this rank's first error, or X264HIP_EPEER */
int  x264hip_shard_stats( x264hip_shard *s, uint64_t *out, int n );
int  x264hip_shard_loopback_verify( x264hip_shard *s, int *n_checked ); /* loopback: every buffer that came back equals what was sent */
void x264hip_shard_close( x264hip_shard *s );                   /* rank 0: also tells the other ranks to leave x264hip_shard_serve */

#ifdef __cplusplus
}
#endif
#endif
