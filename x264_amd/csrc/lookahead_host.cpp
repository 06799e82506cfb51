// lookahead_host.cpp -- host side of the lookahead: the slice-type decision logic of the reference
// (encoder/slicetype.c, encoder/lookahead.c), restated over an evaluation backend.  The decisions stay
// on the CPU ("tiny, stays on host"); every pixel-touching step goes through the
// backend, which in the product is the HIP context of x264hip.hip (no CPU implementation exists here).
//
// Cited line numbers refer to jpsdr/x264 encoder/slicetype.c unless another file is named.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <bitset>
#include <deque>
#include <chrono>
#include <vector>

#include "x264hip.h"

namespace {

enum { T_AUTO = 0, T_IDR = 1, T_I = 2, T_P = 3, T_BREF = 4, T_B = 5, T_KEYFRAME = 6 };
static inline bool is_i( int t ) { return t == T_I || t == T_IDR || t == T_KEYFRAME; }
static inline bool is_b( int t ) { return t == T_B || t == T_BREF; }
static inline bool auto_or_i( int t ) { return t == T_AUTO || is_i( t ); }
static inline bool auto_or_b( int t ) { return t == T_AUTO || is_b( t ); }

const int BMAX = X264HIP_BFRAME_MAX;
const int LOOKAHEAD_MAX = 250; // X264_LOOKAHEAD_MAX, common/base.h:140
const int LA_PREFETCH_CHUNK = 256; // frames of read-ahead per speculative submission (X264HIP_LA_CHUNK overrides it for experiments)
const uint64_t COST_MAX64 = 1ULL << 60;

// adds the wall time of its scope to a statistics slot (x264hip_lookahead_stats)
struct ScopeNs
{
    uint64_t &acc;
    std::chrono::steady_clock::time_point t0;
    explicit ScopeNs( uint64_t &a ) : acc( a ), t0( std::chrono::steady_clock::now() ) {}
    ~ScopeNs() { acc += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>( std::chrono::steady_clock::now() - t0 ).count(); }
};

struct LaFrame
{
    int slot = -1;
    int i_frame = 0;
    int i_type = T_AUTO, i_forced_type = T_AUTO;
    int b_scenecut = 1;      // frame.c:792
    int b_keyframe = 0;
    int i_bframes = 0;
    int refcount = 0;
    int cost_est[BMAX + 2][BMAX + 2];
    int cost_est_aq[BMAX + 2][BMAX + 2];
    int intra_mbs[BMAX + 2];
    bool searched[2][BMAX + 1]; // lowres_mvs[l][d][0][0] != 0x7FFF
    bool intra_calculated = false;
    x264hip_weight weight = { 0, 1, 0, 0 };
    uint64_t pixel_sum = 0, pixel_ssd = 0;
    bool stats_valid = false;
    float weighted_cost_delta[BMAX + 2]; // f_weighted_cost_delta, frame.c:798
    bool prefetch_submitted = false;
    bool weights_prefetched = false;
    int weight_group = 0;             // the speculative submission (flush_prefetch) that queued this frame's weight tests
    // VBV lookahead (slicetype.c:1224-1286): what the frames after this one are planned to be and to cost
    int planned_type[LOOKAHEAD_MAX + 1] = { T_AUTO };
    int planned_satd[LOOKAHEAD_MAX + 1] = { 0 };
    int own_d0 = 0, own_d1 = 0; // the cell the frame is coded with (distances to its references), set by decide()
    int64_t pts = 0;
    int i_duration = 2;         // field units (slicetype.c:1759-1768)
    float f_duration = 0.04f;   // seconds, as MB-tree reads it (:1769-1771)
};

static int ue_size( unsigned v ) // bs_size_ue, common/bitstream.h:278 (2*floor(log2(v+1))+1)
{
    int n = 0;
    for( unsigned t = v + 1; t > 1; t >>= 1 ) n++;
    return 2 * n + 1;
}
static int se_size( int v ) // bs_size_se, common/bitstream.h:291
{
    unsigned t = v <= 0 ? (unsigned)( 1 - 2 * v ) : (unsigned)( 2 * v );
    return ue_size( t - 1 );
}
static inline int clip3i( int v, int lo, int hi ) { return v < lo ? lo : v > hi ? hi : v; }

struct Lookahead
{
    x264hip_la_params p;
    x264hip_backend be;
    x264hip_ctx *ctx = nullptr; // owned when opened on a device
    int i_delay = 0, slicetype_length = 0;
    int b_analyse_keyframe = 0;
    int i_last_keyframe = 0;
    int i_input = 0;
    std::vector<LaFrame *> next;     // h->lookahead->next
    std::deque<LaFrame *> current;   // ofbuf + h->frames.current
    LaFrame *last_nonb = nullptr;
    std::vector<int> free_slots;
    std::vector<LaFrame *> pending_prefetch;
    float f_duration = 0.04f;   // a frame's f_duration at constant frame rate
    int64_t i_prev_duration = 2, i_prev_duration0 = 2; // encoder.c:1644
    uint32_t units_in_tick = 1, time_scale = 50; // sps->vui (encoder/set.c:223-224)
    float qcompress = 0.6f;
    uint64_t stats[8] = { 0 };
    int err = 0;
    x264hip_prefetch_hook prefetch_hook = nullptr; // x264hip_lookahead_open_hooked
    void *prefetch_hook_user = nullptr;
    x264hip_mbtree_hook mbtree_hook = nullptr;     // x264hip_lookahead_set_mbtree_hook
    int chunk_frames = 0;                          // x264hip_lookahead_set_chunk (0 = default)
    int gop_len[3] = { 0, 0, 0 };                  // frames of the last three mini-GOPs decided (newest first): the speculation hint
    int chain_anchor = -1, chain_period = 0;       // the B placement of the latest analysis: its last anchor and the usual anchor spacing
    void *mbtree_hook_user = nullptr;

    int run_mbtree( std::vector<x264hip_mbtree_op> &ops )
    {
        if( mbtree_hook && need( mbtree_hook( mbtree_hook_user, ops.data(), (int)ops.size() ) ) ) return err;
        return need( be.mbtree( be.user, ops.data(), (int)ops.size() ) );
    }

    // ---- frame bookkeeping -------------------------------------------------------------------------
    void release( LaFrame *f )
    {
        if( --f->refcount > 0 ) return;
        free_slots.push_back( f->slot );
        delete f;
    }

    int need( int rc )
    {
        if( rc && !err ) err = rc;
        return rc;
    }

    // ---- slicetype_frame_cost (:836-995) -----------------------------------------------------------
    int frame_cost( LaFrame **frames, int p0, int p1, int b )
    {
        LaFrame *fenc = frames[b];
        stats[0]++;
        if( fenc->cost_est[b - p0][p1 - b] >= 0 )
            return fenc->cost_est[b - p0][p1 - b];
        if( err ) return 0;
        int do_search[2];
        const x264hip_weight *w = nullptr;
        do_search[0] = b != p0 && !fenc->searched[0][b - p0 - 1];
        do_search[1] = b != p1 && !fenc->searched[1][p1 - b - 1];
        if( do_search[0] )
        {
            if( p.weightp && b == p1 )
            {
                weights_analyse( fenc, frames[p0] );
                if( fenc->weight.on ) w = &fenc->weight;
            }
            fenc->searched[0][b - p0 - 1] = true;
        }
        if( do_search[1] ) fenc->searched[1][p1 - b - 1] = true;
        if( err ) return 0;

        x264hip_cost out;
        memset( &out, 0, sizeof( out ) );
        const int with_intra = !fenc->intra_calculated;
        const int ref1_valid = b < p1 && frames[p1]->searched[0][p1 - p0 - 1];
        stats[1]++;
        ScopeNs tm( stats[4] );
        if( need( be.frame_cost( be.user, frames[p0]->slot, frames[p1]->slot, fenc->slot, b - p0, p1 - b, do_search, w, with_intra,
                                 ref1_valid, &out ) ) )
            return 0;
        // The device has answered for a frame of the newest speculative submission: its cells are in, the cost sums of its weight tests
        // were queued right behind them.  The verdicts are taken now and the searches of the pairs that keep a weight go out while the
        // decisions are still busy elsewhere -- a request that brings such a weight then finds the field searched, not merely queued.
        if( !weight_tests.empty() && fenc->weight_group >= weight_tests.front().group )
            speculate_weighted_fields( fenc->weight_group );
        if( b == p1 )
            fenc->intra_mbs[b - p0] = out.intra_mbs;
        if( with_intra )
        {
            fenc->cost_est[0][0] = out.intra_cost_est;
            fenc->cost_est_aq[0][0] = out.intra_cost_est_aq;
        }
        int score;
        if( p0 == p1 )
        {
            // the [0][0] cell: the intra sums when just computed (otherwise the memo above would have hit)
            score = with_intra ? out.intra_cost_est : 0;
            fenc->cost_est_aq[0][0] = with_intra ? out.intra_cost_est_aq : 0;
        }
        else
        {
            score = out.cost_est;
            fenc->cost_est_aq[b - p0][p1 - b] = out.cost_est_aq;
        }
        if( b != p1 )
            score = (int)( (uint64_t)score * 100 / ( 120 + p.dev.bframe_bias ) );
        else
            fenc->intra_calculated = true;
        fenc->cost_est[b - p0][p1 - b] = score;
        return score;
    }

    // ---- x264_weights_analyse, lookahead mode (:284-501 with b_lookahead = 1) ----------------------
    int weight_header_cost( const x264hip_weight &w ) // weight_slice_header_cost (:170-189), luma, one slice
    {
        int denom_cost = ue_size( w.denom ) * 2;
        return p.dev.lambda * ( 10 + denom_cost + 2 * ( se_size( w.scale ) + se_size( w.offset ) ) );
    }

    bool frame_stats( LaFrame *f )
    {
        if( !f->stats_valid )
        {
            if( need( be.frame_stats( be.user, f->slot, &f->pixel_sum, &f->pixel_ssd ) ) ) return false;
            f->stats_valid = true;
        }
        return true;
    }

    // First half of x264_weights_analyse in lookahead mode (:293-330 and the candidate of :401-439): the guessed
    // scale and the one (scale, offset) pair whose cost gets measured.  Pure host arithmetic on the frame totals,
    // so it can also run ahead of time to queue the two cost sums speculatively.  false = no weighting to test.
    bool weight_candidate( LaFrame *fenc, LaFrame *ref, x264hip_weight &guess, x264hip_weight &cand )
    {
        const float epsilon = 1.f / 128.f;
        guess.on = 0; guess.scale = 1; guess.denom = 0; guess.offset = 0;
        if( !frame_stats( fenc ) || !frame_stats( ref ) ) return false;
        const int mb_w = ( p.dev.width + 15 ) / 16, mb_h = ( p.dev.height + 15 ) / 16;
        const int lines = 16 * mb_h, width = 16 * mb_w;
        const int zero_bias = !ref->pixel_ssd;
        float fenc_var = (float)( fenc->pixel_ssd + zero_bias );
        float ref_var = (float)( ref->pixel_ssd + zero_bias );
        float guess_scale = sqrtf( fenc_var / ref_var );
        float fenc_mean = (float)( (uint32_t)fenc->pixel_sum + zero_bias ) / ( lines * width ) / ( 1 << ( p.dev.bit_depth - 8 ) );
        float ref_mean = (float)( (uint32_t)ref->pixel_sum + zero_bias ) / ( lines * width ) / ( 1 << ( p.dev.bit_depth - 8 ) );

        if( fabsf( ref_mean - fenc_mean ) < 0.5f && fabsf( 1.f - guess_scale ) < epsilon )
            return false;
        // weight_get_h264 (:64-75)
        {
            int s = (int)round( guess_scale * 128 );
            guess.offset = 0; guess.denom = 7; guess.scale = s;
            while( guess.denom > 0 && guess.scale > 127 ) { guess.denom--; guess.scale >>= 1; }
            if( guess.scale > 127 ) guess.scale = 127;
        }
        // lookahead mode: one (scale, offset) candidate (:401-439 with both distances 0)
        const int mindenom = guess.denom;
        int cur_scale = clip3i( guess.scale, 0, 127 );
        int cur_offset = (int)( fenc_mean - ref_mean * cur_scale / ( 1 << mindenom ) + 0.5f * 1 );
        if( cur_offset < -128 || cur_offset > 127 )
        {
            cur_offset = clip3i( cur_offset, -128, 127 );
            double v = ( 1 << mindenom ) * ( fenc_mean - cur_offset ) / ref_mean + 0.5f;
            cur_scale = (int)( v < 0 ? 0 : v > 127 ? 127 : v );
        }
        cand.on = 1; cand.scale = cur_scale; cand.denom = mindenom; cand.offset = clip3i( cur_offset, -128, 127 );
        return true;
    }

    // Second half of x264_weights_analyse in lookahead mode (:371-470): the two cost sums, the verdict.  A function of the two pictures
    // alone (the sums come from the backend), so it can also run ahead of the request (speculate_weighted_fields).
    // 1 = the weight `out` is kept, 0 = none, -1 = the backend failed.  ratio: minscore / origscore (X264_WEIGHTP_FAKE).
    int weight_verdict( LaFrame *fenc, LaFrame *ref, const x264hip_weight &guess, const x264hip_weight &cand, x264hip_weight &out, float &ratio )
    {
        int mindenom = guess.denom, minscale = guess.scale, minoff = 0, found = 0;
        unsigned minscore = 0, origscore = 0;
        if( need( be.weight_cost( be.user, fenc->slot, ref->slot, nullptr, &origscore ) ) ) return -1;
        minscore = origscore;
        if( !minscore ) return 0;
        {
            unsigned s = 0;
            if( need( be.weight_cost( be.user, fenc->slot, ref->slot, &cand, &s ) ) ) return -1;
            s += weight_header_cost( cand );
            if( s < minscore ) { minscore = s; minscale = cand.scale; minoff = cand.offset; found = 1; }
        }
        while( mindenom > 0 && !( minscale & 1 ) ) { mindenom--; minscale >>= 1; }
        if( !found || ( minscale == 1 << mindenom && minoff == 0 ) || (float)minscore / origscore > 0.998f )
            return 0;
        out.on = 1; out.scale = minscale; out.denom = mindenom; out.offset = minoff;
        ratio = (float)minscore / origscore;
        return 1;
    }

    void weights_analyse( LaFrame *fenc, LaFrame *ref )
    {
        stats[2]++;
        x264hip_weight &wt = fenc->weight;
        x264hip_weight cand;
        if( !weight_candidate( fenc, ref, wt, cand ) )
        {
            wt.on = 0; wt.scale = 1; wt.denom = 0; wt.offset = 0;
            return;
        }
        if( !fenc->intra_calculated )
        {
            LaFrame *one[1] = { fenc };
            frame_cost( one, 0, 0, 0 );
        }
        if( err ) { wt.on = 0; return; }
        ScopeNs tm( stats[5] );
        // the weights the next requests will arrive at, searched before they are asked for (the sums they need are in by now)
        speculate_weighted_fields( fenc->weight_group );
        const x264hip_weight guess = wt;
        float ratio = 0.f;
        const int v = weight_verdict( fenc, ref, guess, cand, wt, ratio );
        if( v <= 0 )
        {
            wt.on = 0;
            if( v == 0 ) { wt.scale = 1; wt.denom = 0; wt.offset = 0; }
            return;
        }
        stats[3]++;
        if( p.weightp < 0 ) // X264_WEIGHTP_FAKE (:462-463)
            fenc->weighted_cost_delta[fenc->i_frame - ref->i_frame - 1] = ratio;
    }

    // Every (frame, reference) pair whose weight test was queued with the last speculative submissions (flush_prefetch) and has not been
    // decided yet: the verdict is a function of the two pictures, so it is taken now for all of them -- their sums arrived with the ones
    // the current request is about to read -- and the pairs that keep a weight go to the backend as ONE batch of weighted searches
    // (x264hip_prefetch_weighted_fields).  A P request that first-triggers such a field then finds it searched; which fields ARE first
    // triggered by a P request is the caller's business, as ever (a field somebody asks for differently is searched on demand).
    struct WeightTest { int f_frame, r_frame; x264hip_weight guess, cand; int group; };
    std::vector<WeightTest> weight_tests; // (in submission order)
    int weight_group_serial = 0;
    void speculate_weighted_fields( int upto_group )
    {
        if( weight_tests.empty() || !be.prefetch_weighted_fields || err ) { weight_tests.clear(); return; }
        std::vector<WeightTest> todo, later;
        for( const WeightTest &t : weight_tests ) ( t.group <= upto_group ? todo : later ).push_back( t );
        weight_tests.swap( later ); // (tests of a submission the device has not answered for yet wait for their turn)
        if( todo.empty() ) return;
        auto resident = [&]( int number ) -> LaFrame * {
            if( last_nonb && last_nonb->i_frame == number ) return last_nonb;
            for( LaFrame *f : next ) if( f->i_frame == number ) return f;
            return nullptr;
        };
        std::vector<int> sf, sr;
        std::vector<x264hip_weight> ws;
        for( const WeightTest &t : todo )
        {
            LaFrame *f = resident( t.f_frame ), *r = resident( t.r_frame );
            if( !f || !r ) continue;
            x264hip_weight w = t.guess;
            float ratio = 0.f;
            const int v = weight_verdict( f, r, t.guess, t.cand, w, ratio );
            if( v < 0 ) return;
            if( v > 0 ) { sf.push_back( f->slot ); sr.push_back( r->slot ); ws.push_back( w ); }
        }
        if( !sf.empty() )
            need( be.prefetch_weighted_fields( be.user, (int)sf.size(), sf.data(), sr.data(), ws.data() ) );
    }

    // =================================================================================================================
    // Decisions (what x264_slicetype_decide / x264_slicetype_analyse and their helpers compute, slicetype.c:1186-1974).
    //
    // The reference decides with mutually calling routines over character strings ("BBP") and running indices.  Here the same
    // decisions are formulated over three notions:
    //   GopPlan   which frames of a window are anchors (coded as P, or as I where a picture type is forced); the rest are B-frames
    //   mini-GOP  two consecutive anchors and the B-frames between them: the unit of every plan cost and of every MB-tree step
    //   passes    the analysis of a window as a fixed sequence of named passes over it
    // One thing is not free: the ORDER in which frame costs are requested, and every early exit that decides whether a request is
    // made at all.  The first request of a (frame, list, distance) field fixes how it is searched (a P request brings its weight,
    // slicetype.c:855-867; a B cell reads its list-1 reference's vectors only if they exist by then, :629) and every cost is
    // memoised, so the request order is part of the reference's observable behaviour and is reproduced exactly.
    // =================================================================================================================

    struct GopPlan
    {
        int len = 0;                                  // the frames at window positions 1 .. len are planned
        std::bitset<LOOKAHEAD_MAX + 2> anchor, intra; // by window position
        bool is_b( int i ) const { return !anchor[i]; }
        void add_bframes( int n ) { len += n; }
        void add_anchor() { anchor.set( ++len ); }
        int anchor_from( int i ) const // the first anchor at or after position i, 0 if there is none
        {
            while( i <= len && !anchor[i] ) i++;
            return i <= len ? i : 0;
        }
    };

    // The B-frames between two anchors, costed while the running total stays under the budget.  With B-pyramid the middle frame of a run
    // of three or more serves as a reference for the two halves and is costed first, unconditionally (slicetype_path_cost's inner part).
    uint64_t add_bframe_costs( LaFrame **w, int left, int right, uint64_t spent, uint64_t budget )
    {
        auto run = [&]( int from, int to, int r0, int r1 ) {
            for( int b = from; b < to && spent < budget; b++ )
                spent += frame_cost( w, r0, r1, b );
        };
        if( p.b_pyramid && right - left > 2 )
        {
            const int mid = left + ( right - left ) / 2;
            spent += frame_cost( w, left, right, mid );
            run( left + 1, mid, left, mid );
            run( mid + 1, right, mid, right );
        }
        else
            run( left + 1, right, left, right );
        return spent;
    }

    // Cost of coding the window the way a plan says, mini-GOP by mini-GOP from the near end; gives up once the budget is exceeded
    // (the total returned is then only known to be too large).
    uint64_t plan_cost( LaFrame **w, const GopPlan &g, uint64_t budget )
    {
        uint64_t spent = 0;
        for( int left = 0, right = g.anchor_from( 1 ); right; left = right, right = g.anchor_from( right + 1 ) )
        {
            spent += g.intra[right] ? frame_cost( w, right, right, right ) : frame_cost( w, left, right, right );
            if( spent > budget )
                break;
            spent = add_bframe_costs( w, left, right, spent, budget );
        }
        return spent;
    }

    // Picture types the caller has forced, imposed on a candidate plan: a forced I / P makes its position an anchor whatever the plan
    // had there (also inside the part inherited from a shorter plan).  The candidate is admissible if, from the last frame of the
    // inherited part on, every forced B is a B-frame in it (the frame that closes the window is exempt) and no forced I / P was one.
    bool impose_forced_types( LaFrame **w, GopPlan &g, int inherited )
    {
        bool admissible = true;
        for( int i = 1; i <= g.len; i++ )
        {
            const int t = w[i]->i_type;
            if( t == T_AUTO )
                continue;
            const bool judged = i >= inherited;
            if( is_b( t ) )
                admissible = admissible && ( !judged || i == g.len || g.is_b( i ) );
            else
            {
                admissible = admissible && ( !judged || !g.is_b( i ) );
                g.anchor.set( i );
                g.intra[i] = is_i( t );
            }
        }
        return admissible;
    }

    // One step of the trellis over B-frame run lengths (b-adapt 2): the cheapest plan for the first `len` frames is the cheapest
    // among { best plan for len - r - 1 frames, then r B-frames, then an anchor }, r = 0 .. bframes.  An admissible candidate beats any
    // inadmissible one whatever it costs; among equals the cheaper wins, the earlier on a tie.  Each candidate is costed only up
    // to the cost of the best so far.
    void best_plan_for( LaFrame **w, int len, std::vector<GopPlan> &best )
    {
        const int longest_run = p.dev.bframes < len - 1 ? p.dev.bframes : len - 1;
        GopPlan chosen;
        uint64_t chosen_cost = COST_MAX64;
        bool chosen_ok = false;
        for( int r = 0; r <= longest_run; r++ )
        {
            GopPlan cand = best[len - r - 1];
            cand.add_bframes( r );
            cand.add_anchor();
            const bool ok = impose_forced_types( w, cand, len - r - 1 );
            if( !ok && chosen_ok )
                continue;
            if( ok && !chosen_ok )
                chosen_cost = COST_MAX64;
            const uint64_t c = plan_cost( w, cand, chosen_cost );
            if( c < chosen_cost )
            {
                chosen_cost = c; chosen_ok = ok; chosen = cand;
            }
        }
        best[len] = chosen;
    }

    // ---- scene changes (scenecut_internal / scenecut, :1384-1468) -------------------------------------------------------------
    // Is frame p1, predicted from p0, so badly predicted that it should start a GOP?  The inter cost is compared with the intra
    // cost scaled by a bias that grows with the distance from the last key frame (float arithmetic exactly as the reference's).
    bool cut_between( LaFrame **w, int p0, int p1 )
    {
        LaFrame *f = w[p1];
        frame_cost( w, p0, p1, p1 );
        const int intra = f->cost_est[0][0], inter = f->cost_est[p1 - p0][0];
        const int since_key = f->i_frame - i_last_keyframe;
        float hi = p.scenecut_threshold / 100.0;
        float lo = hi * 0.25;
        if( p.keyint_min == p.keyint_max )
            lo = hi;
        float bias;
        if( since_key <= p.keyint_min / 4 || p.intra_refresh )
            bias = lo / 4;
        else if( since_key <= p.keyint_min )
            bias = lo * since_key / p.keyint_min;
        else
            bias = lo + ( hi - lo ) * ( since_key - p.keyint_min ) / ( p.keyint_max - p.keyint_min );
        return inter >= ( 1.0 - bias ) * intra;
    }

    // Scene change at p1?  With look_ahead (the check in front of a new mini-GOP, B-frames enabled) the frames up to `horizon` are
    // examined first, so that a flash or a short burst of changing pictures does not open a GOP: every frame after which the
    // picture returns to what p0 showed is cleared, and so is every frame of the stretch unless the far end is a cut from it.
    bool scene_change( LaFrame **w, int p0, int p1, bool look_ahead, int n_frames, int reach )
    {
        if( look_ahead && p.dev.bframes )
        {
            const int horizon = p0 + 1 + ( p.b_adapt == 2 ? p.dev.bframes : 1 );
            const int far_end = horizon < n_frames ? horizon : n_frames;
            for( int q = p1; q <= far_end; q++ )
                if( !cut_between( w, p0, q ) )
                    for( int k = q; k > p0; k-- )
                        w[k]->b_scenecut = 0;
            for( int q = p0; q <= far_end; q++ )
                if( horizon > reach || ( q < far_end && cut_between( w, q, far_end ) ) )
                    w[q]->b_scenecut = 0;
        }
        return w[p1]->b_scenecut && cut_between( w, p0, p1 );
    }

    // ---- MB-tree: the frame-cost requests keep memoisation / first-trigger state identical to the reference (macroblock_tree,
    // :1091-1184); the propagation itself (mbtree_propagate_cost / _list, macroblock_tree_finish) is recorded as a step list and
    // handed to the backend in one call -- it never feeds back into the decisions.
    static double clip_duration( double f ) { return f < 0.01 ? 0.01 : f > 1.0 ? 1.0 : f; } // CLIP_DURATION, ratecontrol.h:34-40

    void mbt_zero( std::vector<x264hip_mbtree_op> &ops, LaFrame *f )
    {
        x264hip_mbtree_op o;
        memset( &o, 0, sizeof( o ) );
        o.type = X264HIP_MBT_ZERO; o.slot_b = o.slot_p0 = o.slot_p1 = f->slot;
        ops.push_back( o );
    }
    void mbt_simple( std::vector<x264hip_mbtree_op> &ops, int type, LaFrame *a, LaFrame *b )
    {
        x264hip_mbtree_op o;
        memset( &o, 0, sizeof( o ) );
        o.type = type; o.slot_b = a->slot; o.slot_p0 = o.slot_p1 = b->slot;
        ops.push_back( o );
    }
    void mbt_propagate( std::vector<x264hip_mbtree_op> &ops, LaFrame **frames, float average_duration, int p0, int p1, int b, int referenced )
    {
        x264hip_mbtree_op o;
        memset( &o, 0, sizeof( o ) );
        o.type = X264HIP_MBT_PROPAGATE;
        o.slot_b = frames[b]->slot; o.slot_p0 = frames[p0]->slot; o.slot_p1 = frames[p1]->slot;
        o.dist_p0 = b - p0; o.dist_p1 = p1 - b; o.referenced = referenced;
        int dsf = ( ( ( b - p0 ) << 8 ) + ( ( p1 - p0 ) >> 1 ) ) / ( p1 - p0 );
        o.bipred_weight = p.dev.weighted_bipred ? 64 - ( dsf >> 2 ) : 32;
        o.fps_factor = (float)( clip_duration( frames[b]->f_duration ) / ( clip_duration( average_duration ) * 256.0f ) * 0.5f );
        ops.push_back( o );
        if( vbv_lookahead_on() && referenced ) // slicetype.c:1087-1088: VBV rate control reads f_qp_offset of every reference
            mbt_finish( ops, frames[b], average_duration, b == p1 ? b - p0 : 0 );
    }
    bool vbv_lookahead_on() const { return p.vbv && p.rc_lookahead; }
    void mbt_finish( std::vector<x264hip_mbtree_op> &ops, LaFrame *f, float average_duration, int ref0_distance )
    {
        x264hip_mbtree_op o;
        memset( &o, 0, sizeof( o ) );
        o.type = X264HIP_MBT_FINISH; o.slot_b = o.slot_p0 = o.slot_p1 = f->slot;
        o.fps_factor_i = (int)round( clip_duration( average_duration ) / clip_duration( f->f_duration ) * 256 / 0.5f );
        float weightdelta = 0.0;
        if( ref0_distance && f->weighted_cost_delta[ref0_distance - 1] > 0 )
            weightdelta = ( 1.0 - f->weighted_cost_delta[ref0_distance - 1] );
        o.weightdelta = weightdelta;
        o.strength = 5.0f * ( 1.0f - qcompress );
        ops.push_back( o );
    }


    // The window is walked mini-GOP by mini-GOP from its far end; inside a mini-GOP the B-frames come first (far to near, each
    // adding to both of its references), then the anchor that closes it passes its own total on to the anchor that opens it.
    void macroblock_tree( LaFrame **w, int n, int window_starts_intra )
    {
        const int first = window_starts_intra ? 0 : 1; // nearest frame that may take part
        std::vector<x264hip_mbtree_op> ops;
        float total_duration = 0.0;
        for( int j = 0; j <= n; j++ )
            total_duration += w[j]->f_duration;
        const float average_duration = total_duration / ( n + 1 );
        if( window_starts_intra )
            frame_cost( w, 0, 0, 0 );
        int far = n;
        while( far > 0 && is_b( w[far]->i_type ) )
            far--;
        // Without a lookahead (:1112-1124) the accumulators of the frame that starts the window were left behind by the previous
        // call and stand for the future of the far anchor
        const bool carry_over = !p.rc_lookahead;
        if( carry_over )
        {
            if( window_starts_intra )
            {
                mbt_zero( ops, w[0] );
                mbt_simple( ops, X264HIP_MBT_RESET_QP, w[0], w[0] );
                if( be.mbtree && !err ) run_mbtree( ops );
                return;
            }
            mbt_simple( ops, X264HIP_MBT_SWAP, w[far], w[0] );
            mbt_zero( ops, w[0] );
        }
        else
        {
            if( far < first )
                return;
            mbt_zero( ops, w[far] );
        }
        int gap = 0; // B-frames of the mini-GOP handled last
        while( far > first )
        {
            int near = far - 1;
            while( near > 0 && is_b( w[near]->i_type ) )
                near--;
            if( near < first )
                break;
            frame_cost( w, near, far, far );
            mbt_zero( ops, w[near] );
            gap = far - near - 1;
            if( p.b_pyramid && gap > 1 )
            {
                const int mid = near + ( gap + 1 ) / 2;
                frame_cost( w, near, far, mid );
                mbt_zero( ops, w[mid] );
                for( int b = far - 1; b > near; b-- )
                {
                    if( b == mid )
                        continue;
                    const int r0 = b > mid ? mid : near, r1 = b < mid ? mid : far;
                    frame_cost( w, r0, r1, b );
                    mbt_propagate( ops, w, average_duration, r0, r1, b, 0 );
                }
                mbt_propagate( ops, w, average_duration, near, far, mid, 1 );
            }
            else
                for( int b = far - 1; b > near; b-- )
                {
                    frame_cost( w, near, far, b );
                    mbt_propagate( ops, w, average_duration, near, far, b, 0 );
                }
            mbt_propagate( ops, w, average_duration, near, far, far, 1 );
            far = near;
        }
        if( carry_over ) // :1173-1178
        {
            frame_cost( w, 0, far, far );
            mbt_propagate( ops, w, average_duration, 0, far, far, 1 );
            mbt_simple( ops, X264HIP_MBT_SWAP, w[far], w[0] );
        }
        mbt_finish( ops, w[far], average_duration, far );
        if( p.b_pyramid && gap > 1 && !p.vbv ) // :1182-1183
            mbt_finish( ops, w[far + ( gap + 1 ) / 2], average_duration, 0 );
        if( be.mbtree && !ops.empty() && !err )
        {
            ScopeNs tm( stats[6] );
            run_mbtree( ops );
        }
    }

    // ---- the passes of a window analysis (x264_slicetype_analyse, :1473-1743) ---------------------------------------------------

    // b-adapt 2: trellis over the whole window, then the plan's verdict for every frame but the last
    void pass_trellis( LaFrame **w, int n )
    {
        if( n <= 1 )
            return;
        std::vector<GopPlan> best( n + 1 );
        best[1].add_anchor();
        for( int len = 2; len <= n; len++ )
            best_plan_for( w, len, best );
        const GopPlan &g = best[n];
        for( int j = 1; j < n; j++ )
        {
            if( !g.is_b( j ) )
            {
                if( auto_or_b( w[j]->i_type ) ) w[j]->i_type = T_P;
            }
            else if( w[j]->i_type == T_AUTO )
                w[j]->i_type = T_B;
        }
    }

    // b-adapt 1: frame by frame, "P then P" against "B then P" for the undecided frame and its successor, behind the B-frames
    // already decided since the last anchor
    void pass_greedy( LaFrame **w, int n )
    {
        int anchor_at = 0, room = p.dev.bframes;
        for( int j = 1; j < n; j++ )
        {
            if( j - 1 > 0 && is_b( w[j - 1]->i_type ) )
                room--;
            else
            {
                anchor_at = j - 1;
                room = p.dev.bframes;
            }
            if( !room )
            {
                if( auto_or_b( w[j]->i_type ) ) w[j]->i_type = T_P;
                continue;
            }
            if( w[j]->i_type != T_AUTO )
                continue;
            if( is_b( w[j + 1]->i_type ) )
            {
                w[j]->i_type = T_P;
                continue;
            }
            const int behind = j - anchor_at - 1;
            GopPlan as_p, as_b;
            as_p.add_bframes( behind ); as_p.add_anchor(); as_p.add_anchor();
            as_b.add_bframes( behind + 1 ); as_b.add_anchor();
            const uint64_t cost_p = plan_cost( w + anchor_at, as_p, COST_MAX64 );
            const uint64_t cost_b = plan_cost( w + anchor_at, as_b, cost_p );
            w[j]->i_type = cost_b < cost_p ? T_B : T_P;
        }
    }

    // b-adapt 0: as many B-frames as allowed, a forced B-frame behind an undecided frame excepted
    void pass_fixed( LaFrame **w, int n )
    {
        int room = p.dev.bframes;
        for( int j = 1; j < n; j++ )
        {
            if( !room )
            {
                if( auto_or_b( w[j]->i_type ) ) w[j]->i_type = T_P;
            }
            else if( w[j]->i_type == T_AUTO )
                w[j]->i_type = is_b( w[j + 1]->i_type ) ? T_P : T_B;
            room = is_b( w[j]->i_type ) ? room - 1 : p.dev.bframes;
        }
    }

    // the key frame interval (:1680-1731): a frame max-keyint away from the last key frame becomes one -- or rather the last
    // frame before it that may; an I-frame min-keyint away from the last key frame is promoted to IDR (closed GOP)
    void pass_keyframe_interval( LaFrame **w, int n )
    {
        int last_key = i_last_keyframe, candidate = 0;
        for( int j = 1; j <= n; j++ )
        {
            LaFrame *f = w[j];
            int dist = f->i_frame - last_key;
            if( auto_or_i( f->i_forced_type ) && ( p.open_gop || !is_b( w[j - 1]->i_forced_type ) ) )
                candidate = j;
            if( dist >= p.keyint_max )
            {
                if( candidate != 0 && candidate != j )
                {
                    j = candidate;
                    f = w[j];
                    dist = f->i_frame - last_key;
                }
                candidate = 0;
                if( f->i_type != T_IDR )
                    f->i_type = p.open_gop ? T_I : T_IDR;
            }
            if( f->i_type == T_I && dist >= p.keyint_min )
            {
                if( p.open_gop )
                    last_key = f->i_frame;
                else if( f->i_forced_type != T_I )
                    f->i_type = T_IDR;
            }
            if( f->i_type == T_IDR )
            {
                last_key = f->i_frame;
                if( j > 1 && is_b( w[j - 1]->i_type ) )
                    w[j - 1]->i_type = T_P;
            }
        }
    }

    // What VBV rate control is told about the frames after the next anchor (vbv_lookahead, :1224-1286): their types and costs in
    // coded order.  The CPB duration bookkeeping of the reference (calculate_durations) is constant for progressive
    // constant-frame-rate input and stays with the encoder.
    int planned_cost( LaFrame **w, int p0, int p1, int b ) // vbv_frame_cost, :1186-1198
    {
        const int cost = frame_cost( w, p0, p1, b );
        if( !p.dev.aq_mode )
            return cost;
        if( !p.mb_tree )
            return w[b]->cost_est_aq[b - p0][p1 - b];
        int score = 0; // slicetype_frame_cost_recalculate: the cell under the frame's current quantiser offsets
        if( !be.frame_cost_recalculate ) { need( X264HIP_EINVAL ); return 0; }
        if( need( be.frame_cost_recalculate( be.user, w[b]->slot, b - p0, p1 - b, is_b( w[b]->i_type ), &score ) ) ) return 0;
        return score;
    }
    void pass_vbv_plan( LaFrame **w, int n, int window_starts_intra )
    {
        auto next_anchor = [&]( int from, int limit_inclusive ) {
            while( ( limit_inclusive ? from <= n : from < n ) && is_b( w[from]->i_type ) ) from++;
            return from;
        };
        int left = 0, right = next_anchor( 1, 0 );
        LaFrame *owner = w[window_starts_intra ? left : right]; // the frame rate control will be looking at when it reads the plan
        const int owner_at = window_starts_intra ? left : right;
        int k = 0;
        while( right < n )
        {
            if( owner_at != right ) // the anchor itself, unless it is the owner
            {
                const int from = is_i( w[right]->i_type ) ? right : left;
                owner->planned_satd[k] = planned_cost( w, from, right, right );
                owner->planned_type[k] = w[right]->i_type;
                k++;
            }
            for( int b = left + 1; b < right; b++, k++ ) // its B-frames, coded order
            {
                owner->planned_satd[k] = planned_cost( w, left, right, b );
                owner->planned_type[k] = T_B;
            }
            left = right;
            right = next_anchor( right + 1, 1 );
        }
        owner->planned_type[k] = T_AUTO;
    }

    void analyse( int frames_taken_by_keyframe )
    {
        LaFrame *w[LOOKAHEAD_MAX + 3] = { nullptr };
        const int window_starts_intra = !!frames_taken_by_keyframe;
        if( !last_nonb )
            return;
        // the window: the last anchor and what is queued, no further than the lookahead is guaranteed to be filled (b_deterministic)
        int reach = (int)next.size() < LOOKAHEAD_MAX ? (int)next.size() : LOOKAHEAD_MAX;
        if( reach > slicetype_length + 1 - frames_taken_by_keyframe )
            reach = slicetype_length + 1 - frames_taken_by_keyframe;
        w[0] = last_nonb;
        int queued = 0;
        for( ; queued < reach; queued++ )
            w[queued + 1] = next[queued];
        if( !queued )
        {
            if( p.mb_tree ) macroblock_tree( w, 0, window_starts_intra );
            return;
        }
        // frames up to the next forced key frame take part in the GOP decisions; MB-tree with psy-RD, and the VBV plan, see them all
        const int until_keyint = p.keyint_max - w[0]->i_frame + i_last_keyframe - 1;
        const int n_gop = p.intra_refresh ? queued : queued < until_keyint ? queued : until_keyint;
        int n = n_gop;
        if( ( p.psy && p.mb_tree ) || vbv_lookahead_on() )
            n = queued;
        else if( p.open_gop && n < queued )
            n++;
        else if( n == 0 )
        {
            w[1]->i_type = T_I;
            return;
        }
        // pass 1: does the next frame start a new scene?
        if( auto_or_i( w[1]->i_type ) && p.scenecut_threshold && scene_change( w, 0, 1, true, n_gop, reach ) )
        {
            if( w[1]->i_type == T_AUTO ) w[1]->i_type = T_I;
            return;
        }
        // pass 2: forced key frames take their final type; the frame in front of a forced IDR cannot be a B-frame
        for( int j = 1; j <= n; j++ )
            if( w[j]->i_type == T_KEYFRAME )
                w[j]->i_type = p.open_gop ? T_I : T_IDR;
        for( int j = 2; j <= n; j++ )
            if( w[j]->i_type == T_IDR && auto_or_b( w[j - 1]->i_type ) )
                w[j - 1]->i_type = T_P;
        // pass 3: B-frame placement
        int n_settled = n, first_to_reset;
        if( p.dev.bframes )
        {
            if( p.b_adapt == 2 ) pass_trellis( w, n );
            else if( p.b_adapt == 1 ) pass_greedy( w, n );
            else pass_fixed( w, n );
            if( auto_or_b( w[n]->i_type ) )
                w[n]->i_type = T_P;
            // pass 4: a scene change inside the leading run of B-frames ends the run there
            int lead = 0;
            while( lead < n && is_b( w[lead + 1]->i_type ) )
                lead++;
            for( int j = 1; j < lead + 1; j++ )
                if( w[j]->i_forced_type == T_AUTO && auto_or_i( w[j + 1]->i_forced_type ) && p.scenecut_threshold &&
                    scene_change( w, j, j + 1, false, n_gop, reach ) )
                {
                    w[j]->i_type = T_P;
                    n_settled = j;
                    break;
                }
            first_to_reset = window_starts_intra ? 1 : ( lead + 2 < n_settled + 1 ? lead + 2 : n_settled + 1 );
        }
        else
        {
            for( int j = 1; j <= n; j++ )
                if( auto_or_b( w[j]->i_type ) )
                    w[j]->i_type = T_P;
            first_to_reset = !window_starts_intra + 1;
        }
        // what the placement looks like right now, for the backend's speculation (x264hip_gop_hint): the last anchor in front of the
        // window's closing frame (which is an anchor only because the window ends there) and the spacing most anchors have; frames
        // that enter the window later are expected to continue the pattern from there
        if( be.gop_hint && p.dev.bframes )
        {
            int gaps[8], n_gaps = 0, last = -1;
            for( int j = n - 1; j >= 0 && n_gaps < 8; j-- )
                if( j == 0 || !is_b( w[j]->i_type ) )
                {
                    if( last >= 0 ) gaps[n_gaps++] = last - j;
                    else chain_anchor = w[j]->i_frame;
                    last = j;
                }
            chain_period = 0;
            for( int a = 0; a < n_gaps && !chain_period; a++ )
            {
                int same = 0;
                for( int b = 0; b < n_gaps; b++ ) same += gaps[b] == gaps[a];
                if( n_gaps >= 3 && same * 5 >= n_gaps * 3 ) chain_period = gaps[a];
            }
        }
        // pass 5: MB-tree over the types as they stand
        if( p.mb_tree )
            macroblock_tree( w, n < p.keyint_max ? n : p.keyint_max, window_starts_intra );
        // pass 6: the key frame interval
        if( !p.intra_refresh )
            pass_keyframe_interval( w, n );
        // pass 7: the plan for VBV rate control
        if( vbv_lookahead_on() )
            pass_vbv_plan( w, n, window_starts_intra );
        // only the mini-GOP about to be coded keeps its decisions: everything behind it is decided again with more frames in view
        for( int j = first_to_reset; j <= n; j++ )
            w[j]->i_type = w[j]->i_forced_type;
    }

    // ---- x264_slicetype_decide (:1745-1974): the next mini-GOP in coded order, and the costs rate control wants ahead of time ----
    void set_durations()
    {
        // frame durations (:1755-1771): from the time stamps with VFR input (the last queued frame repeats the previous duration),
        // two field units otherwise
        for( size_t i = 0; i < next.size(); i++ )
        {
            LaFrame *f = next[i];
            f->i_duration = !p.vfr_input ? 2 : i + 1 < next.size() ? (int)( 2 * ( next[i + 1]->pts - f->pts ) ) : (int)i_prev_duration;
            i_prev_duration = f->i_duration;
            f->f_duration = (float)( (double)f->i_duration * units_in_tick / time_scale );
        }
    }

    // the final type of one queued frame given how many B-frames / B-references precede it in the mini-GOP; true = it closes the mini-GOP
    bool settle_type( int pos, int &n_b, int &n_bref )
    {
        LaFrame *f = next[pos];
        if( f->i_type == T_BREF && ( ( p.b_pyramid < 2 && n_bref == p.b_pyramid ) || ( p.b_pyramid == 2 && n_bref && p.frame_refs <= n_bref + 3 ) ) )
            f->i_type = T_B; // no room for another B-reference (:1814-1826)
        if( f->i_type == T_KEYFRAME )
            f->i_type = p.open_gop ? T_I : T_IDR;
        const int key_type = p.open_gop && i_last_keyframe >= 0 ? T_I : T_IDR;
        if( ( !p.intra_refresh || f->i_frame == 0 ) && f->i_frame - i_last_keyframe >= p.keyint_max )
        {
            // the key frame interval is up (:1831-1846)
            if( f->i_type == T_AUTO || f->i_type == T_I )
                f->i_type = key_type;
            else if( f->i_type != T_IDR && !( p.open_gop && f->i_type == T_I ) )
                f->i_type = key_type; // a type forced by the caller gives way (the reference warns)
        }
        if( f->i_type == T_I && f->i_frame - i_last_keyframe >= p.keyint_min )
        {
            if( p.open_gop )
            {
                i_last_keyframe = f->i_frame;
                f->b_keyframe = 1;
            }
            else
                f->i_type = T_IDR;
        }
        if( f->i_type == T_IDR )
        {
            i_last_keyframe = f->i_frame;
            f->b_keyframe = 1;
            if( n_b > 0 ) // closed GOP: the B-frame in front of it becomes the P that ends the previous mini-GOP
            {
                n_b--;
                next[n_b]->i_type = T_P;
                return true;
            }
        }
        if( n_b == p.dev.bframes || n_b + 1 >= (int)next.size() )
            if( f->i_type == T_AUTO || is_b( f->i_type ) )
                f->i_type = T_P;
        if( f->i_type == T_BREF )
            n_bref++;
        if( f->i_type == T_AUTO )
            f->i_type = T_B;
        return !is_b( f->i_type );
    }

    void decide()
    {
        if( next.empty() )
            return;
        set_durations();
        if( ( p.dev.bframes && p.b_adapt ) || p.scenecut_threshold || p.mb_tree || vbv_lookahead_on() )
            analyse( 0 );
        int n_b = 0, n_bref = 0;
        while( !settle_type( n_b, n_b, n_bref ) )
            n_b++;
        next[n_b]->i_bframes = n_b;
        gop_len[2] = gop_len[1]; gop_len[1] = gop_len[0];
        gop_len[0] = is_i( next[n_b]->i_type ) ? 0 : n_b + 1; // (a mini-GOP closed by an I frame says nothing about the ones behind it)
        if( p.b_pyramid && n_b > 1 && !n_bref )
        {
            next[( n_b - 1 ) / 2]->i_type = T_BREF; // the middle B-frame of the run becomes a reference
            n_bref++;
        }
        // costs ahead of time for rate control (:1898-1935)
        if( !p.rc_is_cqp )
        {
            LaFrame *w[BMAX + 3];
            w[0] = last_nonb;
            for( int i = 0; i <= n_b; i++ )
                w[i + 1] = next[i];
            const int a = n_b + 1;                                  // the anchor that closes the mini-GOP
            const int from = is_i( next[n_b]->i_type ) ? a : 0;
            frame_cost( w, from, a, a );
            w[a]->own_d0 = a - from; w[a]->own_d1 = 0;
            const bool vbv_rows = ( from != a || n_b ) && p.vbv;
            if( vbv_rows )
                frame_cost( w, a, a, a ); // intra costs for the row sums (:1918-1919; memoised when already there)
            // the cell every B-frame of the mini-GOP is coded with (:1922-1933); VBV needs their row sums now
            int r0 = 0;
            for( int i = 1; i <= n_b; i++ )
            {
                int r1 = a;
                if( w[i]->i_type == T_B )
                    for( r1 = i; w[r1]->i_type == T_B; )
                        r1++;
                w[i]->own_d0 = i - r0; w[i]->own_d1 = r1 - i;
                if( vbv_rows )
                    frame_cost( w, r0, r1, i );
                if( w[i]->i_type == T_BREF )
                    r0 = i;
            }
        }
        // The main-encode weight analysis of a P frame (:1937-1943, b_lookahead = 0) works on the full-resolution planes and
        // stays with the encoder, but its first step is visible in the lookahead's own outputs: when the luma statistics
        // call for a weight test it computes the frame's lowres intra costs if they are still missing (:365-370), which
        // fills i_cost_est[0][0] / i_cost_est_aq[0][0] of P frames no analysis has looked at yet.
        if( p.weightp >= 1 && next[n_b]->i_type == T_P && last_nonb && !next[n_b]->intra_calculated )
        {
            x264hip_weight guess, cand;
            if( weight_candidate( next[n_b], last_nonb, guess, cand ) )
            {
                LaFrame *one[1] = { next[n_b] };
                frame_cost( one, 0, 0, 0 );
            }
        }
        // coded order (:1945-1960): the anchor, the B-references, the plain B-frames
        if( n_b )
        {
            std::vector<LaFrame *> coded( 1, next[n_b] );
            for( int i = 0; i < n_b; i++ ) if( next[i]->i_type == T_BREF ) coded.push_back( next[i] );
            for( int i = 0; i < n_b; i++ ) if( next[i]->i_type != T_BREF ) coded.push_back( next[i] );
            for( int i = 0; i <= n_b; i++ ) next[i] = coded[i];
        }
    }

    // ---- x264_lookahead_get_frames, non-threaded branch (lookahead.c:223-250) -----------------------
    void get_frames()
    {
        if( !current.empty() || next.empty() ) return;
        flush_prefetch();
        decide();
        LaFrame *new_nonb = next[0];
        if( last_nonb ) release( last_nonb );
        last_nonb = new_nonb;
        new_nonb->refcount++;
        int shift = next[0]->i_bframes + 1;
        for( int i = 0; i < shift; i++ ) current.push_back( next[i] );
        next.erase( next.begin(), next.begin() + shift );
        if( b_analyse_keyframe && is_i( last_nonb->i_type ) )
            analyse( shift );
    }

    // Speculative work is submitted for the part of the queue the next decisions can reach (i_delay + 1 frames)
    // plus a chunk of read-ahead; when more frames than that are already queued (batch ingest) the following
    // chunks are submitted while the host is busy deciding on the current one, so device and host overlap.
    void flush_prefetch()
    {
        pending_prefetch.clear();
        if( !be.prefetch ) return;
        // (a hooked lookahead -- the window shard -- works in smaller chunks: a chunk is one round of collectives, and the other ranks
        // should be busy with the next one while rank 0 decides on this one)
        static const int chunk_env = getenv( "X264HIP_LA_CHUNK" ) ? ( atoi( getenv( "X264HIP_LA_CHUNK" ) ) > 2 ? atoi( getenv( "X264HIP_LA_CHUNK" ) ) : 2 ) : 0;
        // read-ahead per submission: 256 frames of 1080p, fewer of larger pictures -- the same amount of device work per submission.  A 4K
        // stream submitted 256 frames at a time spends its first 60 ms in ONE search launch while the host waits, and speculates every
        // class for all of them before the first request has told the context which classes this caller asks for (0.136 s per 250-frame
        // pass against 0.112 s in chunks of 64; 1080p: 22.7 k frames/s at 256 against 21.2 k at 64, profiles/r04_chunk_sweep.txt)
        const long mbs = (long)( ( p.dev.width + 15 ) / 16 ) * ( ( p.dev.height + 15 ) / 16 );
        const int by_area = (int)( (long)LA_PREFETCH_CHUNK * 8160 / ( mbs > 0 ? mbs : 1 ) );
        const int chunk_default = by_area > LA_PREFETCH_CHUNK ? LA_PREFETCH_CHUNK : by_area < 16 ? 16 : by_area;
        const int chunk = chunk_frames ? chunk_frames : chunk_env ? chunk_env : prefetch_hook ? ( chunk_default < 64 ? chunk_default : 64 ) : chunk_default;
        const int reach = (int)next.size() < i_delay + 2 ? (int)next.size() : i_delay + 2;
        int submitted = 0;
        while( submitted < (int)next.size() && next[submitted]->prefetch_submitted ) submitted++;
        // the next chunk goes out while half a chunk of submitted frames is still ahead of the decisions: its kernels
        // then run while the host works through those (results are waited for per batch, not per stream)
        if( submitted >= (int)next.size() || submitted >= reach + chunk / 2 ) return;
        const int from = submitted > reach ? submitted : reach;
        const int upto = (int)next.size() < from + chunk ? (int)next.size() : from + chunk;
        // everything resident up to there: last_nonb + next[0..upto) (pairs further apart than bframes+1 are skipped by the backend)
        std::vector<int> slots, nums;
        if( last_nonb ) { slots.push_back( last_nonb->slot ); nums.push_back( last_nonb->i_frame ); }
        for( int i = 0; i < upto; i++ )
        {
            slots.push_back( next[i]->slot ); nums.push_back( next[i]->i_frame );
            next[i]->prefetch_submitted = true;
        }
        ScopeNs tm( stats[6] );
        if( be.gop_hint && !err )
        {
            // the placement of the latest analysis if there is one (the frames about to be submitted continue it); else three decided
            // mini-GOPs of one length in a row, counted from the last anchor coded
            if( chain_anchor >= 0 && last_nonb && chain_anchor >= last_nonb->i_frame )
                need( be.gop_hint( be.user, chain_anchor, chain_period ) );
            else
            {
                const int period = gop_len[0] > 0 && gop_len[0] == gop_len[1] && gop_len[1] == gop_len[2] ? gop_len[0] : 0;
                need( be.gop_hint( be.user, last_nonb ? last_nonb->i_frame : 0, period ) );
            }
        }
        if( prefetch_hook )
            need( prefetch_hook( prefetch_hook_user, slots.data(), nums.data(), (int)slots.size() ) );
        else
            need( be.prefetch( be.user, slots.data(), nums.data(), (int)slots.size() ) );
        // the two cost sums of every weight test the decisions can ask for (P evaluations over 1..bframes+1 frames):
        // queued behind the searches, answered later without a round trip
        if( be.prefetch_weight_costs && p.weightp && !err )
        {
            std::vector<LaFrame *> res;
            if( last_nonb ) res.push_back( last_nonb );
            for( int i = 0; i < upto; i++ ) res.push_back( next[i] );
            std::vector<int> sf, sr;
            std::vector<x264hip_weight> ws;
            weight_group_serial++;
            for( LaFrame *f : res )
            {
                if( f->weights_prefetched || f == last_nonb ) continue;
                f->weights_prefetched = true;
                f->weight_group = weight_group_serial;
                for( LaFrame *r : res )
                {
                    const int d = f->i_frame - r->i_frame;
                    if( d < 1 || d > p.dev.bframes + 1 ) continue;
                    x264hip_weight guess, cand;
                    if( !weight_candidate( f, r, guess, cand ) ) continue;
                    sf.push_back( f->slot ); sr.push_back( r->slot ); ws.push_back( cand );
                    if( be.prefetch_weighted_fields ) weight_tests.push_back( WeightTest{ f->i_frame, r->i_frame, guess, cand, weight_group_serial } );
                }
            }
            if( !sf.empty() && !err )
                need( be.prefetch_weight_costs( be.user, (int)sf.size(), sf.data(), sr.data(), ws.data() ) );
        }
    }
};

// ---- device backend thunks ---------------------------------------------------------------------------
static int dev_frame_put( void *u, int slot, const void *luma, int stride, int is_device )
{
    return x264hip_frame_put( (x264hip_ctx *)u, slot, luma, stride, is_device, nullptr, nullptr, 0, nullptr );
}
static int dev_frame_stats( void *u, int slot, uint64_t *s, uint64_t *q ) { return x264hip_frame_stats( (x264hip_ctx *)u, slot, s, q ); }
static int dev_weight_cost( void *u, int f, int r, const x264hip_weight *w, unsigned *c ) { return x264hip_weight_cost( (x264hip_ctx *)u, f, r, w, c ); }
static int dev_frame_cost( void *u, int p0, int p1, int b, int d0, int d1, const int ds[2], const x264hip_weight *w, int wi, int rv, x264hip_cost *o )
{
    return x264hip_frame_cost( (x264hip_ctx *)u, p0, p1, b, d0, d1, ds, w, wi, rv, o );
}
static int dev_prefetch( void *u, const int *s, const int *n, int c ) { return x264hip_prefetch( (x264hip_ctx *)u, s, n, c ); }
static int dev_mbtree( void *u, const x264hip_mbtree_op *ops, int n ) { return x264hip_mbtree( (x264hip_ctx *)u, ops, n ); }
static int dev_qp_offsets( void *u, int slot, float *q ) { return x264hip_get_qp_offsets( (x264hip_ctx *)u, slot, q ); }
static int dev_recalc( void *u, int b, int d0, int d1, int aq, int *score ) { return x264hip_frame_cost_recalculate( (x264hip_ctx *)u, b, d0, d1, aq, score ); }
static int dev_row_satds( void *u, int slot, int d0, int d1, int *rows ) { return x264hip_get_lowres_costs( (x264hip_ctx *)u, slot, d0, d1, nullptr, rows ); }
static int dev_frame_put_yuv( void *u, int slot, const void *luma, int stride, const void *cb, const void *cr, int cstride, int is_device )
{
    return x264hip_frame_put( (x264hip_ctx *)u, slot, luma, stride, is_device, cb, cr, cstride, nullptr );
}
static int dev_add_quant_offsets( void *u, int slot, const float *q ) { return x264hip_frame_add_quant_offsets( (x264hip_ctx *)u, slot, q ); }
static int dev_put_batch_yuv( void *u, int n, const int *slots, const void *const *luma, int stride, const void *const *cb, const void *const *cr, int cstride )
{
    return x264hip_frame_put_batch_yuv( (x264hip_ctx *)u, n, slots, luma, stride, cb, cr, cstride );
}
static int dev_put_batch( void *u, int n, const int *slots, const void *const *luma, int stride )
{
    return x264hip_frame_put_batch( (x264hip_ctx *)u, n, slots, luma, stride );
}
static int dev_gop_hint( void *u, int anchor, int period ) { return x264hip_gop_hint( (x264hip_ctx *)u, anchor, period ); }
static int dev_flush( void *u ) { return x264hip_flush( (x264hip_ctx *)u ); }
static int dev_prefetch_weighted_fields( void *u, int n, const int *sf, const int *sr, const x264hip_weight *w )
{
    return x264hip_prefetch_weighted_fields( (x264hip_ctx *)u, n, sf, sr, w );
}
static int dev_prefetch_weights( void *u, int n, const int *sf, const int *sr, const x264hip_weight *w )
{
    return x264hip_prefetch_weight_costs( (x264hip_ctx *)u, n, sf, sr, w );
}

} // namespace

struct x264hip_lookahead
{
    Lookahead L;
};

static int la_init( x264hip_lookahead *la, const x264hip_la_params *params )
{
    Lookahead &L = la->L;
    L.p = *params;
    const x264hip_la_params &p = L.p;
    if( p.dev.bframes < 0 || p.dev.bframes > BMAX || p.keyint_max < 1 || p.rc_lookahead < 0 || p.rc_lookahead > LOOKAHEAD_MAX ||
        p.b_adapt < 0 || p.b_adapt > 2 || p.b_pyramid < 0 || p.b_pyramid > 2 )
        return X264HIP_EINVAL;

    // encoder.c:1601-1612 with one frame thread, no lookahead thread, cfr input
    if( p.b_adapt == 2 )
        L.i_delay = ( p.dev.bframes > 3 ? p.dev.bframes : 3 ) * 4;
    else
        L.i_delay = p.dev.bframes;
    if( p.mb_tree || p.vbv )
        L.i_delay = L.i_delay > p.rc_lookahead ? L.i_delay : p.rc_lookahead;
    L.slicetype_length = L.i_delay;
    L.i_delay += !!p.vfr_input; // encoder.c:1612: one more frame until the duration of the first is known
    L.b_analyse_keyframe = p.mb_tree || ( p.vbv && p.rc_lookahead ); // lookahead.c:140-141 (no stats read)
    {
        // slicetype.c:1767-1771: i_duration = 2 field units for a progressive frame, time base 1/(2*fps)
        const int fn = p.fps_num > 0 ? p.fps_num : 25, fd = p.fps_den > 0 ? p.fps_den : 1;
        L.f_duration = (float)( (double)2 * fd / ( 2.0 * fn ) );
        // timebase (encoder.c:1119-1123, :1559-1560) and the VUI timing it becomes (set.c:223-224); encoder.c:1644
        auto gcd = []( uint64_t a, uint64_t b ) { while( b ) { uint64_t t = a % b; a = b; b = t; } return a; };
        uint64_t rfn = fn, rfd = fd, g = gcd( rfn, rfd );
        rfn /= g; rfd /= g;
        uint64_t tn = p.vfr_input && p.timebase_num > 0 && p.timebase_den > 0 ? p.timebase_num : rfd;
        uint64_t td = p.vfr_input && p.timebase_num > 0 && p.timebase_den > 0 ? p.timebase_den : rfn;
        g = gcd( tn, td ); tn /= g; td /= g;
        if( td * 2 > 0xFFFFFFFFull ) return X264HIP_EINVAL;
        L.units_in_tick = (uint32_t)tn; L.time_scale = (uint32_t)( td * 2 );
        L.i_prev_duration = L.i_prev_duration0 = (int64_t)( ( rfd * L.time_scale ) / ( rfn * L.units_in_tick ) );
        L.qcompress = p.qcompress > 0 ? p.qcompress : 0.6f;
    }
    L.i_last_keyframe = -p.keyint_max;
    return X264HIP_OK;
}

static int slots_needed( const x264hip_la_params *p )
{
    int delay = p->b_adapt == 2 ? ( p->dev.bframes > 3 ? p->dev.bframes : 3 ) * 4 : p->dev.bframes;
    if( ( p->mb_tree || p->vbv ) && p->rc_lookahead > delay ) delay = p->rc_lookahead;
    return delay + p->dev.bframes + 8 + !!p->vfr_input;
}

extern "C" int x264hip_lookahead_open_backend( x264hip_lookahead **out, const x264hip_la_params *params, const x264hip_backend *backend )
{
    if( !out || !params || !backend || !backend->frame_put || !backend->frame_cost || !backend->frame_stats || !backend->weight_cost )
        return X264HIP_EINVAL;
    x264hip_lookahead *la = new x264hip_lookahead();
    int rc = la_init( la, params );
    if( rc ) { delete la; return rc; }
    la->L.be = *backend;
    int n = params->dev.max_frames > 0 ? params->dev.max_frames : slots_needed( params );
    for( int i = 0; i < n; i++ ) la->L.free_slots.push_back( i ); // (taken from the front: the first picture gets slot 0)
    *out = la;
    return X264HIP_OK;
}

/* The cell and field classes this flow can ask for under a configuration (what x264hip_lookahead_open states through
 * x264hip_spec_classes; pure host arithmetic).  With B-pyramid every walk over a run of B-frames -- add_bframe_costs, mbtree_ops and the
 * costs ahead of time -- splits the run at the same middle frame: between anchors `len` apart the middle frame sees
 * ( len / 2, len - len / 2 ) and every other B-frame two references inside one half.  Everything else -- most of the triangle for long
 * runs -- need never be speculated.  Without B-pyramid every B cell of the triangle can be asked for.  (Picture types forced by the
 * caller can put a B-reference somewhere else; the cells that follow from it are then evaluated on demand, x264hip_spec_classes.) */
extern "C" int x264hip_lookahead_classes( const x264hip_la_params *params, unsigned char *cell_allowed, unsigned *mask_l0, unsigned *mask_l1 )
{
    if( !params || !cell_allowed || !mask_l0 || !mask_l1 || params->dev.bframes < 0 || params->dev.bframes > X264HIP_BFRAME_MAX ) return X264HIP_EINVAL;
    const int bf = params->dev.bframes, ns = bf + 2;
    memset( cell_allowed, 0, (size_t)ns * ns );
    unsigned l0 = 0, l1 = 0;
    auto allow = [&]( int d0, int d1 ) {
        cell_allowed[d0 * ns + d1] = 1;
        if( d0 ) l0 |= 1u << ( d0 - 1 );
        if( d1 ) l1 |= 1u << ( d1 - 1 );
    };
    allow( 0, 0 );
    for( int d = 1; d <= bf + 1; d++ ) allow( d, 0 );
    for( int len = 2; len <= bf + 1; len++ )
    {
        if( params->b_pyramid && len > 2 )
        {
            const int h0 = len / 2, h1 = len - len / 2;
            allow( h0, h1 );
            for( int k = 1; k < h0; k++ ) allow( k, h0 - k );
            for( int k = 1; k < h1; k++ ) allow( k, h1 - k );
        }
        else
            for( int k = 1; k < len; k++ ) allow( k, len - k );
    }
    *mask_l0 = l0; *mask_l1 = l1;
    return X264HIP_OK;
}

extern "C" int x264hip_lookahead_open( x264hip_lookahead **out, int device, const x264hip_la_params *params )
{
    if( !out || !params ) return X264HIP_EINVAL;
    x264hip_la_params p = *params;
    if( p.dev.max_frames <= 0 ) p.dev.max_frames = slots_needed( &p );
    p.dev.no_edges = !( p.mb_tree || p.vbv ); // slicetype.c:823: the evaluations visit the edge blocks only for MB-tree and VBV
    x264hip_ctx *ctx = nullptr;
    int rc = x264hip_open( &ctx, device, &p.dev );
    if( rc ) return rc;
    x264hip_backend be = { ctx, dev_frame_put, dev_frame_stats, dev_weight_cost, dev_frame_cost, dev_prefetch, dev_mbtree, dev_qp_offsets, dev_put_batch, dev_prefetch_weights, dev_recalc, dev_row_satds, dev_frame_put_yuv, dev_add_quant_offsets, dev_put_batch_yuv, dev_gop_hint, dev_flush, dev_prefetch_weighted_fields };
    rc = x264hip_lookahead_open_backend( out, &p, &be );
    if( rc ) { x264hip_close( ctx ); return rc; }
    ( *out )->L.ctx = ctx;
    {
        const int ns = p.dev.bframes + 2;
        std::vector<unsigned char> ok( (size_t)ns * ns );
        unsigned l0 = 0, l1 = 0;
        x264hip_lookahead_classes( &p, ok.data(), &l0, &l1 );
        x264hip_spec_classes( ctx, ok.data(), l0, l1 );
    }
    return X264HIP_OK;
}

extern "C" int x264hip_lookahead_open_hooked( x264hip_lookahead **out, int device, const x264hip_la_params *params, x264hip_prefetch_hook hook, void *user )
{
    int rc = x264hip_lookahead_open( out, device, params );
    if( rc ) return rc;
    ( *out )->L.prefetch_hook = hook;
    ( *out )->L.prefetch_hook_user = user;
    return X264HIP_OK;
}

extern "C" int x264hip_lookahead_set_chunk( x264hip_lookahead *la, int frames )
{
    if( !la || frames < 0 ) return X264HIP_EINVAL;
    la->L.chunk_frames = frames == 0 ? 0 : frames < 2 ? 2 : frames;
    return X264HIP_OK;
}

extern "C" int x264hip_lookahead_set_mbtree_hook( x264hip_lookahead *la, x264hip_mbtree_hook hook, void *user )
{
    if( !la ) return X264HIP_EINVAL;
    la->L.mbtree_hook = hook;
    la->L.mbtree_hook_user = user;
    return X264HIP_OK;
}

extern "C" void x264hip_lookahead_close( x264hip_lookahead *la )
{
    if( !la ) return;
    Lookahead &L = la->L;
    for( auto f : L.next ) delete f;
    for( auto f : L.current ) if( f != L.last_nonb ) delete f;
    if( L.last_nonb ) delete L.last_nonb;
    if( L.ctx ) x264hip_close( L.ctx );
    delete la;
}

/* forget every frame (end of a sequence); the device context and its allocations are kept */
extern "C" int x264hip_lookahead_reset( x264hip_lookahead *la )
{
    if( !la ) return X264HIP_EINVAL;
    Lookahead &L = la->L;
    if( L.err ) return L.err;
    for( auto f : L.next ) { L.free_slots.push_back( f->slot ); delete f; }
    for( auto f : L.current )
        if( f != L.last_nonb ) { L.free_slots.push_back( f->slot ); delete f; }
    if( L.last_nonb ) { L.free_slots.push_back( L.last_nonb->slot ); delete L.last_nonb; }
    L.next.clear(); L.current.clear(); L.last_nonb = nullptr; L.pending_prefetch.clear();
    L.chain_anchor = -1; L.chain_period = 0;
    // (gop_len stays: the next sequence on this context is expected to be decided like the last one until it shows otherwise)
    L.i_input = 0;
    L.i_last_keyframe = -L.p.keyint_max;
    L.i_prev_duration = L.i_prev_duration0;
    return X264HIP_OK;
}

extern "C" int x264hip_lookahead_delayed_frames( x264hip_lookahead *la )
{
    return la ? (int)( la->L.next.size() + la->L.current.size() ) : X264HIP_EINVAL;
}

extern "C" x264hip_ctx *x264hip_lookahead_ctx( x264hip_lookahead *la ) { return la ? la->L.ctx : nullptr; }
extern "C" int x264hip_lookahead_delay( x264hip_lookahead *la ) { return la ? la->L.i_delay : X264HIP_EINVAL; }

static LaFrame *new_frame( Lookahead &L, int forced_type )
{
    LaFrame *f = new LaFrame();
    // the slot that has been free the longest: what MB-tree launch last worked on it is long over when its next picture arrives (the
    // backend orders an ingest behind the launch that names its slot, x264hip.hip mbt_guard_slot)
    f->slot = L.free_slots.front(); L.free_slots.erase( L.free_slots.begin() );
    f->i_frame = L.i_input++;
    // an unknown picture type is taken as AUTO (x264_frame_copy_picture, frame.c:392-400)
    f->i_forced_type = f->i_type = forced_type < T_AUTO || forced_type > T_KEYFRAME ? T_AUTO : forced_type;
    f->pts = f->i_frame;
    f->f_duration = L.f_duration;
    f->refcount = 1;
    // encoder.c:1617-1625 b_have_lowres: constant QP without any analysis has no lowres planes, and the -1 marks are the work of
    // x264_frame_init_lowres (mc.c:473): without it the cells keep the zeros of the frame's allocation
    const bool have_lowres = !L.p.rc_is_cqp || L.p.b_adapt || L.p.scenecut_threshold || L.p.mb_tree || L.p.weightp;
    memset( f->cost_est, have_lowres ? -1 : 0, sizeof( f->cost_est ) );
    memset( f->cost_est_aq, 0, sizeof( f->cost_est_aq ) );
    memset( f->intra_mbs, 0, sizeof( f->intra_mbs ) );
    memset( f->searched, 0, sizeof( f->searched ) );       // mc.c:479-481
    memset( f->weighted_cost_delta, 0, sizeof( f->weighted_cost_delta ) );
    return f;
}

extern "C" int x264hip_lookahead_put_frames( x264hip_lookahead *la, int n, const void *const *luma_dev, int stride )
{
    return x264hip_lookahead_put_pictures( la, n, luma_dev, stride, nullptr, nullptr, 0, nullptr, nullptr );
}

// A whole clip of device-resident frames through the lookahead in one call: every frame put, every decided frame taken, in the order
// x264_encoder_encode would (paced: one put, one get per frame, then the flush; otherwise all frames put first -- deep read-ahead).
// The same sequence of x264hip_lookahead_put_frame(s) / x264hip_lookahead_get_frame calls a caller would make, without a trip through
// the caller's language per frame.
extern "C" int x264hip_lookahead_run_frames( x264hip_lookahead *la, int n, const void *const *luma_dev, int stride, int paced, x264hip_la_frame *out,
                                             int *n_out )
{
    if( !la || n <= 0 || !luma_dev || !out || !n_out ) return X264HIP_EINVAL;
    int got = 0, m = 0, rc = X264HIP_OK;
    *n_out = 0;
    if( !paced )
    {
        rc = x264hip_lookahead_put_frames( la, n, luma_dev, stride );
        if( rc ) return rc;
    }
    else
        for( int i = 0; i < n; i++ )
        {
            rc = x264hip_lookahead_put_frame( la, luma_dev[i], stride, 1, 0 );
            if( rc ) return rc;
            rc = x264hip_lookahead_get_frame_ex( la, 0, out + m, &got, nullptr );
            if( rc ) return rc;
            m += got;
        }
    while( m < n )
    {
        rc = x264hip_lookahead_get_frame_ex( la, 1, out + m, &got, nullptr );
        if( rc ) return rc;
        if( !got ) break;
        m++;
    }
    *n_out = m;
    return X264HIP_OK;
}

extern "C" int x264hip_lookahead_put_pictures( x264hip_lookahead *la, int n, const void *const *luma_dev, int stride, const void *const *cb_dev,
                                               const void *const *cr_dev, int cstride, const int *types, const int64_t *pts )
{
    if( !la || n <= 0 || !luma_dev || ( !cb_dev ) != ( !cr_dev ) ) return X264HIP_EINVAL;
    Lookahead &L = la->L;
    ScopeNs tm_api( L.stats[7] );
    if( L.err ) return L.err;
    if( cb_dev ? !L.be.frame_put_batch_yuv : !L.be.frame_put_batch )
    {
        for( int i = 0; i < n; i++ )
        {
            const void *planes[3] = { luma_dev[i], cb_dev ? cb_dev[i] : nullptr, cb_dev ? cr_dev[i] : nullptr };
            const int strides[3] = { stride, cstride, cstride };
            int rc = x264hip_lookahead_put_picture( la, planes, strides, 1, types ? types[i] : T_AUTO, pts ? pts[i] : L.i_input );
            if( rc ) return rc;
        }
        return X264HIP_OK;
    }
    if( (int)L.free_slots.size() < n ) return X264HIP_ESTATE;
    std::vector<LaFrame *> fr;
    std::vector<int> slots;
    for( int i = 0; i < n; i++ )
    {
        fr.push_back( new_frame( L, types ? types[i] : T_AUTO ) );
        fr.back()->pts = pts ? pts[i] : fr.back()->i_frame;
        slots.push_back( fr.back()->slot );
    }
    int rc = cb_dev ? L.be.frame_put_batch_yuv( L.be.user, n, slots.data(), luma_dev, stride, cb_dev, cr_dev, cstride )
                    : L.be.frame_put_batch( L.be.user, n, slots.data(), luma_dev, stride );
    if( rc )
    {
        for( auto f : fr ) { L.free_slots.push_back( f->slot ); delete f; }
        L.i_input -= n;
        return L.need( rc );
    }
    for( auto f : fr ) { L.next.push_back( f ); L.pending_prefetch.push_back( f ); }
    return X264HIP_OK;
}

extern "C" int x264hip_lookahead_put_frame( x264hip_lookahead *la, const void *luma, int stride, int is_device, int forced_type )
{
    // x264_picture_t.i_pts defaults to the frame number here (what the CLI gives constant-frame-rate input)
    return x264hip_lookahead_put_frame_pts( la, luma, stride, is_device, forced_type, la ? la->L.i_input : 0 );
}

extern "C" int x264hip_lookahead_put_frame_pts( x264hip_lookahead *la, const void *luma, int stride, int is_device, int forced_type, int64_t pts )
{
    const void *planes[3] = { luma, nullptr, nullptr };
    const int strides[3] = { stride, 0, 0 };
    return x264hip_lookahead_put_picture( la, planes, strides, is_device, forced_type, pts );
}

extern "C" int x264hip_lookahead_put_picture( x264hip_lookahead *la, const void *const planes[3], const int strides[3], int is_device, int forced_type, int64_t pts )
{
    if( !planes || !strides ) return X264HIP_EINVAL;
    x264hip_picture pic;
    for( int k = 0; k < 3; k++ ) { pic.planes[k] = planes[k]; pic.strides[k] = strides[k]; }
    pic.is_device = is_device; pic.i_type = forced_type; pic.i_pts = pts; pic.quant_offsets = nullptr;
    return x264hip_lookahead_put( la, &pic );
}

extern "C" int x264hip_lookahead_put( x264hip_lookahead *la, const x264hip_picture *pic )
{
    if( !la || !pic ) return X264HIP_EINVAL;
    const void *const *planes = pic->planes;
    const int *strides = pic->strides;
    const int is_device = pic->is_device, forced_type = pic->i_type;
    const int64_t pts = pic->i_pts;
    if( !planes[0] || ( !planes[1] ) != ( !planes[2] ) || ( pic->quant_offsets && !la->L.be.add_quant_offsets ) ) return X264HIP_EINVAL;
    const void *luma = planes[0];
    const int stride = strides[0];
    const bool with_chroma = planes[1] != nullptr;
    if( with_chroma && ( !la->L.be.frame_put_yuv || strides[1] != strides[2] ) ) return X264HIP_EINVAL;
    Lookahead &L = la->L;
    ScopeNs tm_api( L.stats[7] );
    if( L.err ) return L.err;
    if( L.free_slots.empty() ) return X264HIP_ESTATE;
    LaFrame *f = new_frame( L, forced_type );
    f->pts = pts;
    f->f_duration = L.f_duration;
    int rc = with_chroma ? L.be.frame_put_yuv( L.be.user, f->slot, luma, stride, planes[1], planes[2], strides[1], is_device )
                         : L.be.frame_put( L.be.user, f->slot, luma, stride, is_device );
    if( !rc && pic->quant_offsets )
        rc = L.be.add_quant_offsets( L.be.user, f->slot, pic->quant_offsets );
    if( rc )
    {
        L.free_slots.push_back( f->slot );
        delete f;
        L.i_input--;
        return L.need( rc );
    }
    L.next.push_back( f );
    L.pending_prefetch.push_back( f );
    // (X264HIP_LA_PUT_FLUSH=1, measured in round 6 and left off: the speculative work for a picture submitted WITH the picture instead of at
    //  the next decision.  Its searches then start a put earlier -- and every launch is a chain of W + 2 (H - 1) block searches on ONE
    //  stream, one after the other: three launches per mini-GOP instead of one, 1 290 against 3 160 frames/s, profiles/r06_paced.txt)
    static const bool put_flush = getenv( "X264HIP_LA_PUT_FLUSH" ) && atoi( getenv( "X264HIP_LA_PUT_FLUSH" ) ) == 1;
    if( put_flush && !L.prefetch_hook && L.i_input > L.i_delay && !L.err ) // (a hooked lookahead -- the window shard -- keeps its chunks: a chunk is a round of collectives)
    {
        L.flush_prefetch();
        if( L.err ) return L.err;
    }
    return X264HIP_OK;
}

extern "C" int x264hip_lookahead_get_frame( x264hip_lookahead *la, int flush, x264hip_la_frame *out, int *got )
{
    return x264hip_lookahead_get_frame_ex( la, flush, out, got, nullptr );
}

extern "C" int x264hip_lookahead_get_frame_ex( x264hip_lookahead *la, int flush, x264hip_la_frame *out, int *got, float *qp_offset )
{
    return x264hip_lookahead_get_frame_vbv( la, flush, out, got, qp_offset, nullptr, nullptr, nullptr );
}

extern "C" int x264hip_lookahead_get_frame_vbv( x264hip_lookahead *la, int flush, x264hip_la_frame *out, int *got, float *qp_offset,
                                                 x264hip_la_vbv *vbv, int *row_satds, int *row_satds_intra )
{
    if( !la || !out || !got ) return X264HIP_EINVAL;
    Lookahead &L = la->L;
    // the backend entries the extra outputs need are checked before a frame is taken off the queue
    if( ( vbv && L.p.mb_tree && !L.be.frame_cost_recalculate ) || ( ( row_satds || row_satds_intra ) && !L.be.get_row_satds ) )
        return X264HIP_EINVAL;
    ScopeNs tm_api( L.stats[7] );
    *got = 0;
    if( L.err ) return L.err;
    // encoder.c:3428-3433: nothing to encode while the lookahead delay fills
    if( !flush && L.i_input <= L.i_delay + 1 - 1 )
        return X264HIP_OK;
    if( L.current.empty() )
        L.get_frames();
    if( L.err ) return L.err;
    if( L.current.empty() )
        return X264HIP_OK;
    LaFrame *f = L.current.front();
    L.current.pop_front();
    // the frame has left the queue: whatever happens below (a backend error while fetching the extra outputs), its slot goes back
    struct Release { Lookahead &L; LaFrame *f; ~Release() { L.release( f ); } } release_on_exit{ L, f };
    out->frame = f->i_frame; out->type = f->i_type; out->bframes = f->i_bframes; out->keyframe = f->b_keyframe;
    for( int i = 0; i < BMAX + 2; i++ )
    {
        for( int j = 0; j < BMAX + 2; j++ )
        {
            bool alloc = i <= L.p.dev.bframes + 1 && j <= L.p.dev.bframes + 1;
            out->cost_est[i][j] = alloc ? f->cost_est[i][j] : -1;
            out->cost_est_aq[i][j] = alloc ? f->cost_est_aq[i][j] : 0;
        }
        out->intra_mbs[i] = f->intra_mbs[i];
    }
    *got = 1;
    // the last delayed frame of a flush: nothing the analysis asked the backend for may stay queued behind the end of the stream
    if( flush && L.current.empty() && L.next.empty() && L.be.flush && L.need( L.be.flush( L.be.user ) ) )
        return L.err;
    if( qp_offset && L.be.get_qp_offsets && ( L.p.mb_tree || L.p.dev.aq_mode ) ) // the arrays exist with AQ on (frame.c:217-226)
        if( L.need( L.be.get_qp_offsets( L.be.user, f->slot, qp_offset ) ) )
            return L.err;
    if( vbv )
    {
        vbv->n_planned = 0;
        if( !is_b( f->i_type ) )
            while( vbv->n_planned < LOOKAHEAD_MAX && f->planned_type[vbv->n_planned] != T_AUTO ) vbv->n_planned++;
        memcpy( vbv->planned_type, f->planned_type, sizeof( vbv->planned_type ) );
        memcpy( vbv->planned_satd, f->planned_satd, sizeof( vbv->planned_satd ) );
        if( is_b( f->i_type ) ) vbv->planned_type[0] = T_AUTO;
        vbv->dist_p0 = f->own_d0; vbv->dist_p1 = f->own_d1;
        // x264_rc_analyse_slice (slicetype.c:1976-2009), the part without intra refresh: the frame's cost as rate control
        // takes it, and with MB-tree the row sums rewritten under the final quantiser offsets
        // (rate control analyses B frames only with VBV, ratecontrol.c:2472-2474; without it their cell may never have been evaluated)
        int cost = is_b( f->i_type ) && !L.p.vbv ? -1 : f->cost_est[f->own_d0][f->own_d1];
        if( cost >= 0 )
        {
            if( L.p.mb_tree )
            {
                if( L.need( L.be.frame_cost_recalculate( L.be.user, f->slot, f->own_d0, f->own_d1, is_b( f->i_type ), &cost ) ) ) return L.err;
                int unused = 0;
                if( f->own_d0 && L.p.vbv && f->cost_est[0][0] >= 0 )
                    if( L.need( L.be.frame_cost_recalculate( L.be.user, f->slot, 0, 0, is_b( f->i_type ), &unused ) ) ) return L.err;
            }
            else if( L.p.dev.aq_mode )
                cost = f->cost_est_aq[f->own_d0][f->own_d1];
        }
        vbv->satd = cost;
    }
    if( row_satds && f->cost_est[f->own_d0][f->own_d1] >= 0 )
        if( L.need( L.be.get_row_satds( L.be.user, f->slot, f->own_d0, f->own_d1, row_satds ) ) ) return L.err;
    if( row_satds_intra && f->cost_est[0][0] >= 0 ) // computed by any evaluation that found them missing, B evaluations included (slicetype.c:714-757)
        if( L.need( L.be.get_row_satds( L.be.user, f->slot, 0, 0, row_satds_intra ) ) ) return L.err;
    return X264HIP_OK;
}

extern "C" int x264hip_lookahead_stats( x264hip_lookahead *la, uint64_t *out, int n )
{
    if( !la || !out ) return X264HIP_EINVAL;
    for( int i = 0; i < n && i < 8; i++ ) out[i] = la->L.stats[i];
    return X264HIP_OK;
}
