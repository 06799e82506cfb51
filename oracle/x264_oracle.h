/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the x264 lookahead / ME hot path.
 *
 * Plain C restatement of the algorithms in the reference (jpsdr/x264 @ /root/reference):
 * common/pixel.c, common/mc.c, common/predict.c, common/dct.c, common/quant.c, encoder/me.c and
 * encoder/slicetype.c (each function below cites the file:line it follows).  It is the checker the
 * HIP path is compared with; nothing in the product (x264_amd/) links, loads or calls it.
 *
 * PARITY PIN: tests/test_oracle_vs_ref.py checks every function here bit-exactly against the real
 * reference built by oracle/build_ref.sh (oracle/_ref/libx264ref{8,10}.so), and
 * tests/golden/ holds fixtures generated from that build for boxes without /root/reference.
 *
 * The file is compiled once per bit depth; symbols are prefixed or8_ / or10_.
 */
#ifndef X264_ORACLE_H
#define X264_ORACLE_H
#include <stdint.h>

#define OR_PAD 32          /* lowres border, reference PADH/PADV (common/frame.h:32-33) */
#define OR_FENC_STRIDE 16  /* common/common.h FENC_STRIDE */
#define OR_FDEC_STRIDE 32  /* common/common.h FDEC_STRIDE */

enum { OR_ME_DIA = 0, OR_ME_HEX = 1 };

/* lookahead configuration (what lowres_context_init + x264_param_t give the reference) */
typedef struct or_la_cfg
{
    int mb_w, mb_h;         /* 8x8 lowres blocks per row / column (= 16x16 MBs of the full frame) */
    int stride;             /* lowres plane stride in pixels */
    int lambda;             /* x264_lambda_tab[X264_LOOKAHEAD_QP] */
    int me_method;          /* OR_ME_DIA / OR_ME_HEX (slicetype.c:50-59) */
    int subpel_refine;      /* lookahead h->mb.i_subpel_refine: 2 or 4 */
    int me_range;           /* param.analyse.i_me_range */
    int mv_range;           /* param.analyse.i_mv_range */
    int subme;              /* param.analyse.i_subpel_refine (user): intra extra modes, bidir path */
    int mbcmp_satd;         /* mbcmp = SATD (subme > 1) else SAD (encoder.c:1409-1427) */
    int fpelcmp_satd;       /* fpelcmp = SATD only for me=tesa */
    int weighted_bipred;
    int aq_mode;
    int bframe_bias;
    int n_slices;           /* param.i_lookahead_threads: horizontal bands searched independently (slicetype.c:917-918) */
    int do_edges;           /* slicetype.c:823: edge blocks are evaluated only with MB-tree / VBV / frames <= 2 blocks wide or high */
    const uint16_t *cost_mv;    /* centred table, index range +-(2*4*mv_range) */
} or_la_cfg;

typedef struct or_weight
{
    int on, scale, denom, offset;
} or_weight;

/* per-evaluation summary as slicetype_frame_cost leaves it (before the B *100/(120+bias) scaling) */
typedef struct or_cell_out
{
    int cost_est, cost_est_aq, intra_mbs;
    int intra_cost_est, intra_cost_est_aq; /* the [0][0] cell if intra was computed in this call */
} or_cell_out;

/* one call of the main-encode motion search (x264_me_t + the h->mb limits it reads), see x264_oracle.c */
#define OR_ME_FULL(D, PIX) \
typedef struct or##D##_me_full \
{ \
    int i_pixel;                /* PIXEL_16x16 .. PIXEL_4x4 (0..6) */ \
    int me_method;              /* 0 dia, 1 hex, 2 umh, 3 esa, 4 tesa */ \
    int subpel_refine;          /* h->mb.i_subpel_refine */ \
    int me_range; \
    int mbcmp_satd, fpelcmp_satd; \
    const PIX *fenc;            /* FENC layout, stride 16 */ \
    const PIX *ref[4];          /* full / H / V / HV planes at the block origin */ \
    int stride; \
    const uint16_t *integral;   /* 8x8-sum plane at the block origin (TESA only) */ \
    long integral_lower;        /* elements from there to the 4x4-sum plane */ \
    int mvp[2]; \
    int lim_min[2], lim_max[2]; /* h->mb.mv_limit_fpel */ \
    int spel_min[2], spel_max[2]; \
    const uint16_t *cost_mv;    /* centred */ \
} or##D##_me_full;
OR_ME_FULL( 8, uint8_t )
OR_ME_FULL( 10, uint16_t )

#define OR_DECL(D, PIX, COEF, UCOEF) \
void or##D##_lowres_init( const PIX *src, int src_stride, int width, int height, int mb_w, int mb_h, \
                          PIX *p0, PIX *ph, PIX *pv, PIX *pc, int stride ); \
void or##D##_lowres_core( const PIX *src, PIX *d0, PIX *dh, PIX *dv, PIX *dc, int src_stride, int dst_stride, int w, int h ); \
void or##D##_integral_init4h( uint16_t *sum, const PIX *pix, long stride ); \
void or##D##_integral_init8h( uint16_t *sum, const PIX *pix, long stride ); \
void or##D##_integral_init4v( uint16_t *sum8, uint16_t *sum4, long stride ); \
void or##D##_integral_init8v( uint16_t *sum8, long stride ); \
int  or##D##_ads( int n_dc, const int *enc_dc, const uint16_t *sums, int delta, const uint16_t *cost_mvx, int16_t *mvs, int width, int thresh ); \
void or##D##_me_search_full( const or##D##_me_full *p, const int16_t (*mvc)[2], int n_mvc, int out[4] ); \
int  or##D##_frame_cost_recalculate( int mb_w, int mb_h, const uint16_t *lowres_costs, const float *qp_offset, int *row_satds ); \
void or##D##_hpel_filter( PIX *dsth, PIX *dstv, PIX *dstc, const PIX *src, long stride, int width, int height, int16_t *buf ); \
int  or##D##_sad( const PIX *a, int sa, const PIX *b, int sb, int w, int h ); \
int  or##D##_ssd( const PIX *a, int sa, const PIX *b, int sb, int w, int h ); \
int  or##D##_satd( const PIX *a, int sa, const PIX *b, int sb, int w, int h ); \
int  or##D##_sa8d( const PIX *a, int sa, const PIX *b, int sb, int w ); \
uint64_t or##D##_var( const PIX *a, int sa, int w, int h ); \
int  or##D##_var2( const PIX *fenc, const PIX *fdec, int h, int ssd[2] ); \
uint64_t or##D##_hadamard_ac( const PIX *pix, int stride, int w, int h ); \
int  or##D##_vsad( const PIX *src, long stride, int height ); \
int  or##D##_asd8( const PIX *a, long sa, const PIX *b, long sb, int height ); \
void or##D##_predict_8x8c( int mode, PIX *src ); \
void or##D##_predict_8x8_filter( const PIX *src, PIX *edge ); \
void or##D##_predict_8x8( int mode, PIX *src, const PIX *edge ); \
void or##D##_intra_x3_8x8c( int satd, const PIX *fenc, PIX *fdec, int res[3] ); \
void or##D##_mc_luma( PIX *dst, int ds, const PIX *const planes[4], int stride, int mvx, int mvy, int w, int h, const or_weight *wt ); \
void or##D##_avg( PIX *dst, int ds, const PIX *a, int sa, const PIX *b, int sb, int w, int h, int weight ); \
void or##D##_weight_plane( PIX *dst, const PIX *src, int stride, int width, int lines, const or_weight *wt ); \
unsigned or##D##_weight_cost( const or_la_cfg *c, const PIX *fenc0, const PIX *ref0, const or_weight *wt, const uint16_t *intra_cost ); \
void or##D##_dct( int kind, COEF *out, const PIX *fenc, const PIX *fdec ); \
int  or##D##_quant( int kind, COEF *coef, const UCOEF *mf, const UCOEF *bias, int mf_dc, int bias_dc ); \
void or##D##_intra_costs( const or_la_cfg *c, const PIX *fenc0, uint16_t *intra_cost ); \
void or##D##_search_field( const or_la_cfg *c, const PIX *fenc0, const PIX *const ref[4], const PIX *ref_w, \
                           const or_weight *wt, int16_t (*mvs)[2], int *mv_costs ); \
void or##D##_cell( const or_la_cfg *c, const PIX *fenc0, const PIX *const ref0[4], const PIX *const ref1[4], \
                   int b_bidir, int dist_scale_factor, const or_weight *wt, \
                   const int16_t (*mvs0)[2], const int *costs0, const int16_t (*mvs1)[2], const int *costs1, \
                   const int16_t (*ref1_l0_mvs)[2], const uint16_t *intra_cost, const uint16_t *inv_qscale, \
                   int with_intra, uint16_t *lowres_costs, int *row_satds, int *row_satds_intra, or_cell_out *out ); \
uint64_t or##D##_aq_frame( const PIX *luma, int stride, int width, int height, int mb_w, int mb_h, \
                           const PIX *cb, const PIX *cr, int cstride, int aq_mode, float aq_strength, \
                           uint16_t *inv_qscale, float *qp_offset, uint64_t *ssd_out ); \
uint64_t or##D##_aq_frame_fmt( const PIX *luma, int stride, int width, int height, int mb_w, int mb_h, \
                               const PIX *cb, const PIX *cr, int cstride, int aq_mode, float aq_strength, \
                               uint16_t *inv_qscale, float *qp_offset, uint64_t *ssd_out, int chroma_format, const float *quant_offsets );

OR_DECL( 8, uint8_t, int16_t, uint16_t )
OR_DECL( 10, uint16_t, int32_t, uint32_t )

/* depth independent */
void or_mbtree_propagate( int mb_w, int mb_h, const uint16_t *intra_cost, const uint16_t *lowres_costs, const uint16_t *inv_qscale,
                          const uint16_t *propagate_in, const int16_t (*mvs0)[2], const int16_t (*mvs1)[2],
                          uint16_t *ref0_costs, uint16_t *ref1_costs, int bipred_weight, float fps_factor );
void or_mbtree_finish( int n_mb, const uint16_t *intra_cost, const uint16_t *inv_qscale, const uint16_t *propagate_cost,
                       const float *qp_offset_aq, float *qp_offset, int fps_factor, float weightdelta, float strength );
void or_cost_mv_table( uint16_t *out_centre, int n, int lambda ); /* analyse.c:143-202 */
int  or_lambda_for_depth( int bit_depth );

#endif
