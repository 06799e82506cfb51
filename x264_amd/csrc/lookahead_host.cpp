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
        int mindenom = wt.denom, minscale = wt.scale, minoff = 0, found = 0;
        if( !fenc->intra_calculated )
        {
            LaFrame *one[1] = { fenc };
            frame_cost( one, 0, 0, 0 );
        }
        if( err ) { wt.on = 0; return; }
        unsigned minscore = 0, origscore = 0;
        ScopeNs tm( stats[5] );
        if( need( be.weight_cost( be.user, fenc->slot, ref->slot, nullptr, &origscore ) ) ) { wt.on = 0; return; }
        minscore = origscore;
        if( !minscore ) { wt.on = 0; wt.scale = 1; wt.denom = 0; wt.offset = 0; /* keeps the guessed values off */ return; }
        {
            const int cur_scale = cand.scale, i_off = cand.offset;
            unsigned s = 0;
            if( need( be.weight_cost( be.user, fenc->slot, ref->slot, &cand, &s ) ) ) { wt.on = 0; return; }
            s += weight_header_cost( cand );
            if( s < minscore ) { minscore = s; minscale = cur_scale; minoff = i_off; found = 1; }
        }
        while( mindenom > 0 && !( minscale & 1 ) ) { mindenom--; minscale >>= 1; }
        if( !found || ( minscale == 1 << mindenom && minoff == 0 ) || (float)minscore / origscore > 0.998f )
        {
            wt.on = 0; wt.scale = 1; wt.denom = 0; wt.offset = 0;
            return;
        }
        wt.on = 1; wt.scale = minscale; wt.denom = mindenom; wt.offset = minoff;
        stats[3]++;
        if( p.weightp < 0 ) // X264_WEIGHTP_FAKE (:462-463)
            fenc->weighted_cost_delta[fenc->i_frame - ref->i_frame - 1] = (float)minscore / origscore;
    }

    // ---- slicetype_path_cost (:1288-1327) ----------------------------------------------------------
    uint64_t path_cost( LaFrame **frames, const char *path, uint64_t threshold )
    {
        uint64_t cost = 0;
        int loc = 1, cur_nonb = 0;
        path--; // path[1] describes frames[1]
        while( path[loc] )
        {
            int next_nonb = loc;
            while( path[next_nonb] == 'B' ) next_nonb++;
            if( path[next_nonb] == 'P' )
                cost += frame_cost( frames, cur_nonb, next_nonb, next_nonb );
            else
                cost += frame_cost( frames, next_nonb, next_nonb, next_nonb );
            if( cost > threshold ) break;
            if( p.b_pyramid && next_nonb - cur_nonb > 2 )
            {
                int middle = cur_nonb + ( next_nonb - cur_nonb ) / 2;
                cost += frame_cost( frames, cur_nonb, next_nonb, middle );
                for( int nb = loc; nb < middle && cost < threshold; nb++ )
                    cost += frame_cost( frames, cur_nonb, middle, nb );
                for( int nb = middle + 1; nb < next_nonb && cost < threshold; nb++ )
                    cost += frame_cost( frames, middle, next_nonb, nb );
            }
            else
                for( int nb = loc; nb < next_nonb && cost < threshold; nb++ )
                    cost += frame_cost( frames, cur_nonb, next_nonb, nb );
            loc = next_nonb + 1;
            cur_nonb = next_nonb;
        }
        return cost;
    }

    // ---- slicetype_path: one Viterbi step (:1333-1382) ---------------------------------------------
    void slicetype_path( LaFrame **frames, int length, char ( *best_paths )[LOOKAHEAD_MAX + 1] )
    {
        char paths[2][LOOKAHEAD_MAX + 1];
        int num_paths = p.dev.bframes + 1 < length ? p.dev.bframes + 1 : length;
        uint64_t best_cost = COST_MAX64;
        int best_possible = 0, idx = 0;
        for( int path = 0; path < num_paths; path++ )
        {
            int len = length - ( path + 1 );
            memcpy( paths[idx], best_paths[len % ( BMAX + 1 )], len );
            memset( paths[idx] + len, 'B', path );
            strcpy( paths[idx] + len + path, "P" );
            int possible = 1;
            for( int i = 1; i <= length; i++ )
            {
                int t = frames[i]->i_type;
                if( t == T_AUTO ) continue;
                if( is_b( t ) )
                    possible = possible && ( i < len || i == length || paths[idx][i - 1] == 'B' );
                else
                {
                    possible = possible && ( i < len || paths[idx][i - 1] != 'B' );
                    paths[idx][i - 1] = is_i( t ) ? 'I' : 'P';
                }
            }
            if( possible || !best_possible )
            {
                if( possible && !best_possible ) best_cost = COST_MAX64;
                uint64_t cost = path_cost( frames, paths[idx], best_cost );
                if( cost < best_cost )
                {
                    best_cost = cost; best_possible = possible; idx ^= 1;
                }
            }
        }
        memcpy( best_paths[length % ( BMAX + 1 )], paths[idx ^ 1], length );
    }

    // ---- scenecut (:1384-1468) ---------------------------------------------------------------------
    int scenecut_internal( LaFrame **frames, int p0, int p1 )
    {
        LaFrame *frame = frames[p1];
        frame_cost( frames, p0, p1, p1 );
        int icost = frame->cost_est[0][0];
        int pcost = frame->cost_est[p1 - p0][0];
        float f_bias;
        int gop_size = frame->i_frame - i_last_keyframe;
        float thresh_max = p.scenecut_threshold / 100.0;
        float thresh_min = thresh_max * 0.25;
        if( p.keyint_min == p.keyint_max ) thresh_min = thresh_max;
        if( gop_size <= p.keyint_min / 4 || p.intra_refresh )
            f_bias = thresh_min / 4;
        else if( gop_size <= p.keyint_min )
            f_bias = thresh_min * gop_size / p.keyint_min;
        else
            f_bias = thresh_min + ( thresh_max - thresh_min ) * ( gop_size - p.keyint_min ) / ( p.keyint_max - p.keyint_min );
        return pcost >= ( 1.0 - f_bias ) * icost;
    }

    int scenecut( LaFrame **frames, int p0, int p1, int real_scenecut, int num_frames, int i_max_search )
    {
        if( real_scenecut && p.dev.bframes )
        {
            int origmaxp1 = p0 + 1;
            if( p.b_adapt == 2 ) origmaxp1 += p.dev.bframes;
            else origmaxp1++;
            int maxp1 = origmaxp1 < num_frames ? origmaxp1 : num_frames;
            for( int curp1 = p1; curp1 <= maxp1; curp1++ )
                if( !scenecut_internal( frames, p0, curp1 ) )
                    for( int i = curp1; i > p0; i-- )
                        frames[i]->b_scenecut = 0;
            for( int curp0 = p0; curp0 <= maxp1; curp0++ )
                if( origmaxp1 > i_max_search || ( curp0 < maxp1 && scenecut_internal( frames, curp0, maxp1 ) ) )
                    frames[curp0]->b_scenecut = 0;
        }
        if( !frames[p1]->b_scenecut ) return 0;
        return scenecut_internal( frames, p0, p1 );
    }

    // ---- macroblock_tree (:1091-1184).  The frame-cost evaluations keep memoisation / first-trigger state
    // identical to the reference; the propagation itself (mbtree_propagate_cost/list, macroblock_tree_finish) is
    // recorded as a step list and handed to the backend in one call (it never feeds back into the decisions).
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

    void macroblock_tree( LaFrame **frames, int num_frames, int b_intra )
    {
        int idx = !b_intra, last_nonb, cur_nonb = 1, bframes = 0;
        std::vector<x264hip_mbtree_op> ops;
        float total_duration = 0.0;
        for( int j = 0; j <= num_frames; j++ )
            total_duration += frames[j]->f_duration;
        float average_duration = total_duration / ( num_frames + 1 );
        int i = num_frames;
        if( b_intra ) frame_cost( frames, 0, 0, 0 );
        while( i > 0 && is_b( frames[i]->i_type ) ) i--;
        last_nonb = i;
        // Lookahead-less MB-tree (:1112-1124): the accumulators of the frame that starts the window were left behind by the
        // previous call and serve as the extrapolated future of frames[last_nonb]
        const bool lookaheadless = !p.rc_lookahead;
        if( lookaheadless )
        {
            if( b_intra )
            {
                mbt_zero( ops, frames[0] );
                mbt_simple( ops, X264HIP_MBT_RESET_QP, frames[0], frames[0] );
                if( be.mbtree && !err ) need( be.mbtree( be.user, ops.data(), (int)ops.size() ) );
                return;
            }
            mbt_simple( ops, X264HIP_MBT_SWAP, frames[last_nonb], frames[0] );
            mbt_zero( ops, frames[0] );
        }
        else
        {
            if( last_nonb < idx ) return;
            mbt_zero( ops, frames[last_nonb] );
        }
        while( i-- > idx )
        {
            cur_nonb = i;
            while( is_b( frames[cur_nonb]->i_type ) && cur_nonb > 0 ) cur_nonb--;
            if( cur_nonb < idx ) break;
            frame_cost( frames, cur_nonb, last_nonb, last_nonb );
            mbt_zero( ops, frames[cur_nonb] );
            bframes = last_nonb - cur_nonb - 1;
            if( p.b_pyramid && bframes > 1 )
            {
                int middle = ( bframes + 1 ) / 2 + cur_nonb;
                frame_cost( frames, cur_nonb, last_nonb, middle );
                mbt_zero( ops, frames[middle] );
                while( i > cur_nonb )
                {
                    int q0 = i > middle ? middle : cur_nonb;
                    int q1 = i < middle ? middle : last_nonb;
                    if( i != middle )
                    {
                        frame_cost( frames, q0, q1, i );
                        mbt_propagate( ops, frames, average_duration, q0, q1, i, 0 );
                    }
                    i--;
                }
                mbt_propagate( ops, frames, average_duration, cur_nonb, last_nonb, middle, 1 );
            }
            else
                while( i > cur_nonb )
                {
                    frame_cost( frames, cur_nonb, last_nonb, i );
                    mbt_propagate( ops, frames, average_duration, cur_nonb, last_nonb, i, 0 );
                    i--;
                }
            mbt_propagate( ops, frames, average_duration, cur_nonb, last_nonb, last_nonb, 1 );
            last_nonb = cur_nonb;
        }
        if( lookaheadless ) // :1173-1178
        {
            frame_cost( frames, 0, last_nonb, last_nonb );
            mbt_propagate( ops, frames, average_duration, 0, last_nonb, last_nonb, 1 );
            mbt_simple( ops, X264HIP_MBT_SWAP, frames[last_nonb], frames[0] );
        }
        mbt_finish( ops, frames[last_nonb], average_duration, last_nonb );
        if( p.b_pyramid && bframes > 1 && !p.vbv ) // :1182-1183
            mbt_finish( ops, frames[last_nonb + ( bframes + 1 ) / 2], average_duration, 0 );
        if( be.mbtree && !ops.empty() && !err )
        {
            ScopeNs tm( stats[6] );
            need( be.mbtree( be.user, ops.data(), (int)ops.size() ) );
        }
    }

    // ---- x264_slicetype_analyse (:1473-1743) -------------------------------------------------------
    void analyse( int intra_minigop )
    {
        LaFrame *frames[LOOKAHEAD_MAX + 3] = { nullptr };
        int num_frames, orig_num_frames, keyint_limit, framecnt;
        int i_max_search = (int)next.size() < LOOKAHEAD_MAX ? (int)next.size() : LOOKAHEAD_MAX;
        if( i_max_search > slicetype_length + 1 - intra_minigop ) // b_deterministic
            i_max_search = slicetype_length + 1 - intra_minigop;
        int keyframe = !!intra_minigop;
        if( !last_nonb ) return;
        frames[0] = last_nonb;
        for( framecnt = 0; framecnt < i_max_search; framecnt++ )
            frames[framecnt + 1] = next[framecnt];
        if( !framecnt )
        {
            if( p.mb_tree ) macroblock_tree( frames, 0, keyframe );
            return;
        }
        keyint_limit = p.keyint_max - frames[0]->i_frame + i_last_keyframe - 1;
        orig_num_frames = num_frames = p.intra_refresh ? framecnt : framecnt < keyint_limit ? framecnt : keyint_limit;
        if( ( p.psy && p.mb_tree ) || vbv_lookahead_on() )
            num_frames = framecnt;
        else if( p.open_gop && num_frames < framecnt )
            num_frames++;
        else if( num_frames == 0 )
        {
            frames[1]->i_type = T_I;
            return;
        }
        if( auto_or_i( frames[1]->i_type ) && p.scenecut_threshold && scenecut( frames, 0, 1, 1, orig_num_frames, i_max_search ) )
        {
            if( frames[1]->i_type == T_AUTO ) frames[1]->i_type = T_I;
            return;
        }
        for( int j = 1; j <= num_frames; j++ )
            if( frames[j]->i_type == T_KEYFRAME )
                frames[j]->i_type = p.open_gop ? T_I : T_IDR;
        for( int j = 2; j <= num_frames; j++ )
            if( frames[j]->i_type == T_IDR && auto_or_b( frames[j - 1]->i_type ) )
                frames[j - 1]->i_type = T_P;

        int num_analysed_frames = num_frames, reset_start;
        const int bf = p.dev.bframes;
        if( bf )
        {
            if( p.b_adapt == 2 )
            {
                if( num_frames > 1 )
                {
                    static thread_local char best_paths[BMAX + 1][LOOKAHEAD_MAX + 1];
                    memset( best_paths, 0, sizeof( best_paths ) );
                    strcpy( best_paths[1], "P" );
                    int best_path_index = num_frames % ( BMAX + 1 );
                    for( int j = 2; j <= num_frames; j++ )
                        slicetype_path( frames, j, best_paths );
                    for( int j = 1; j < num_frames; j++ )
                    {
                        if( best_paths[best_path_index][j - 1] != 'B' )
                        {
                            if( auto_or_b( frames[j]->i_type ) ) frames[j]->i_type = T_P;
                        }
                        else if( frames[j]->i_type == T_AUTO )
                            frames[j]->i_type = T_B;
                    }
                }
            }
            else if( p.b_adapt == 1 )
            {
                int last_nb = 0, num_b = bf;
                char path[LOOKAHEAD_MAX + 1];
                for( int j = 1; j < num_frames; j++ )
                {
                    if( j - 1 > 0 && is_b( frames[j - 1]->i_type ) )
                        num_b--;
                    else
                    {
                        last_nb = j - 1;
                        num_b = bf;
                    }
                    if( !num_b )
                    {
                        if( auto_or_b( frames[j]->i_type ) ) frames[j]->i_type = T_P;
                        continue;
                    }
                    if( frames[j]->i_type != T_AUTO ) continue;
                    if( is_b( frames[j + 1]->i_type ) )
                    {
                        frames[j]->i_type = T_P;
                        continue;
                    }
                    int nb = j - last_nb - 1;
                    memset( path, 'B', nb );
                    strcpy( path + nb, "PP" );
                    uint64_t cost_p = path_cost( frames + last_nb, path, COST_MAX64 );
                    strcpy( path + nb, "BP" );
                    uint64_t cost_b = path_cost( frames + last_nb, path, cost_p );
                    frames[j]->i_type = cost_b < cost_p ? T_B : T_P;
                }
            }
            else
            {
                int num_b = bf;
                for( int j = 1; j < num_frames; j++ )
                {
                    if( !num_b )
                    {
                        if( auto_or_b( frames[j]->i_type ) ) frames[j]->i_type = T_P;
                    }
                    else if( frames[j]->i_type == T_AUTO )
                        frames[j]->i_type = is_b( frames[j + 1]->i_type ) ? T_P : T_B;
                    if( is_b( frames[j]->i_type ) ) num_b--;
                    else num_b = bf;
                }
            }
            if( auto_or_b( frames[num_frames]->i_type ) )
                frames[num_frames]->i_type = T_P;
            int num_b = 0;
            while( num_b < num_frames && is_b( frames[num_b + 1]->i_type ) ) num_b++;
            for( int j = 1; j < num_b + 1; j++ )
                if( frames[j]->i_forced_type == T_AUTO && auto_or_i( frames[j + 1]->i_forced_type ) && p.scenecut_threshold &&
                    scenecut( frames, j, j + 1, 0, orig_num_frames, i_max_search ) )
                {
                    frames[j]->i_type = T_P;
                    num_analysed_frames = j;
                    break;
                }
            reset_start = keyframe ? 1 : ( num_b + 2 < num_analysed_frames + 1 ? num_b + 2 : num_analysed_frames + 1 );
        }
        else
        {
            for( int j = 1; j <= num_frames; j++ )
                if( auto_or_b( frames[j]->i_type ) ) frames[j]->i_type = T_P;
            reset_start = !keyframe + 1;
        }
        if( p.mb_tree )
            macroblock_tree( frames, num_frames < p.keyint_max ? num_frames : p.keyint_max, keyframe );

        // keyframe limit (:1680-1731), not with intra refresh
        if( !p.intra_refresh )
        {
            int last_keyframe = i_last_keyframe, last_possible = 0;
            for( int j = 1; j <= num_frames; j++ )
            {
                LaFrame *frm = frames[j];
                int keyframe_dist = frm->i_frame - last_keyframe;
                if( auto_or_i( frm->i_forced_type ) )
                    if( p.open_gop || !is_b( frames[j - 1]->i_forced_type ) )
                        last_possible = j;
                if( keyframe_dist >= p.keyint_max )
                {
                    if( last_possible != 0 && last_possible != j )
                    {
                        j = last_possible;
                        frm = frames[j];
                        keyframe_dist = frm->i_frame - last_keyframe;
                    }
                    last_possible = 0;
                    if( frm->i_type != T_IDR ) frm->i_type = p.open_gop ? T_I : T_IDR;
                }
                if( frm->i_type == T_I && keyframe_dist >= p.keyint_min )
                {
                    if( p.open_gop )
                        last_keyframe = frm->i_frame;
                    else if( frm->i_forced_type != T_I )
                        frm->i_type = T_IDR;
                }
                if( frm->i_type == T_IDR )
                {
                    last_keyframe = frm->i_frame;
                    if( j > 1 && is_b( frames[j - 1]->i_type ) ) frames[j - 1]->i_type = T_P;
                }
            }
        }
        if( vbv_lookahead_on() )
            vbv_lookahead( frames, num_frames, keyframe );
        for( int j = reset_start; j <= num_frames; j++ )
            frames[j]->i_type = frames[j]->i_forced_type;
    }

    // ---- vbv_frame_cost / vbv_lookahead (:1186-1198, :1224-1286) ------------------------------------
    // The planned types and costs of the frames after the next non-B frame, in coded order, for VBV rate control
    // (ratecontrol.c:2290-2320).  The CPB duration bookkeeping of the reference (calculate_durations) is constant
    // for progressive constant-frame-rate input and stays with the encoder.
    int vbv_frame_cost( LaFrame **frames, int p0, int p1, int b )
    {
        int cost = frame_cost( frames, p0, p1, b );
        if( p.dev.aq_mode )
        {
            if( p.mb_tree )
            {
                // slicetype_frame_cost_recalculate: the cell under the frame's current quantiser offsets
                int score = 0;
                if( !be.frame_cost_recalculate ) { need( X264HIP_EINVAL ); return 0; }
                if( need( be.frame_cost_recalculate( be.user, frames[b]->slot, b - p0, p1 - b, is_b( frames[b]->i_type ), &score ) ) ) return 0;
                return score;
            }
            return frames[b]->cost_est_aq[b - p0][p1 - b];
        }
        return cost;
    }
    void vbv_lookahead( LaFrame **frames, int num_frames, int keyframe )
    {
        int last_nonb = 0, cur_nonb = 1, idx = 0;
        while( cur_nonb < num_frames && is_b( frames[cur_nonb]->i_type ) ) cur_nonb++;
        int next_nonb = keyframe ? last_nonb : cur_nonb;
        LaFrame *dst = frames[next_nonb];
        while( cur_nonb < num_frames )
        {
            if( next_nonb != cur_nonb ) // P/I cost: not the cost of next_nonb itself
            {
                int p0 = is_i( frames[cur_nonb]->i_type ) ? cur_nonb : last_nonb;
                dst->planned_satd[idx] = vbv_frame_cost( frames, p0, cur_nonb, cur_nonb );
                dst->planned_type[idx] = frames[cur_nonb]->i_type;
                idx++;
            }
            for( int i = last_nonb + 1; i < cur_nonb; i++, idx++ ) // the B-frames, coded order
            {
                dst->planned_satd[idx] = vbv_frame_cost( frames, last_nonb, cur_nonb, i );
                dst->planned_type[idx] = T_B;
            }
            last_nonb = cur_nonb;
            cur_nonb++;
            while( cur_nonb <= num_frames && is_b( frames[cur_nonb]->i_type ) ) cur_nonb++;
        }
        dst->planned_type[idx] = T_AUTO;
    }

    // ---- x264_slicetype_decide (:1745-1974), type logic and the final cost evaluations --------------
    void decide()
    {
        if( next.empty() ) return;
        // frame durations (:1755-1771): from the time stamps with VFR input (the last queued frame repeats the previous duration),
        // two field units otherwise
        for( size_t i = 0; i < next.size(); i++ )
        {
            if( p.vfr_input )
                next[i]->i_duration = i + 1 < next.size() ? (int)( 2 * ( next[i + 1]->pts - next[i]->pts ) ) : (int)i_prev_duration;
            else
                next[i]->i_duration = 2;
            i_prev_duration = next[i]->i_duration;
            next[i]->f_duration = (float)( (double)next[i]->i_duration * units_in_tick / time_scale );
        }
        if( ( p.dev.bframes && p.b_adapt ) || p.scenecut_threshold || p.mb_tree || vbv_lookahead_on() )
            analyse( 0 );
        int bframes, brefs;
        LaFrame *frm;
        for( bframes = 0, brefs = 0;; bframes++ )
        {
            frm = next[bframes];
            if( frm->i_type == T_BREF && p.b_pyramid < 2 && brefs == p.b_pyramid )
                frm->i_type = T_B;
            else if( frm->i_type == T_BREF && p.b_pyramid == 2 && brefs && p.frame_refs <= ( brefs + 3 ) )
                frm->i_type = T_B;
            if( frm->i_type == T_KEYFRAME )
                frm->i_type = p.open_gop ? T_I : T_IDR;
            if( ( !p.intra_refresh || frm->i_frame == 0 ) && frm->i_frame - i_last_keyframe >= p.keyint_max )
            {
                if( frm->i_type == T_AUTO || frm->i_type == T_I )
                    frm->i_type = p.open_gop && i_last_keyframe >= 0 ? T_I : T_IDR;
                int warn = frm->i_type != T_IDR;
                if( warn && p.open_gop ) warn &= frm->i_type != T_I;
                if( warn )
                    frm->i_type = p.open_gop && i_last_keyframe >= 0 ? T_I : T_IDR;
            }
            if( frm->i_type == T_I && frm->i_frame - i_last_keyframe >= p.keyint_min )
            {
                if( p.open_gop )
                {
                    i_last_keyframe = frm->i_frame;
                    frm->b_keyframe = 1;
                }
                else
                    frm->i_type = T_IDR;
            }
            if( frm->i_type == T_IDR )
            {
                i_last_keyframe = frm->i_frame;
                frm->b_keyframe = 1;
                if( bframes > 0 )
                {
                    bframes--;
                    next[bframes]->i_type = T_P;
                }
            }
            if( bframes == p.dev.bframes || bframes + 1 >= (int)next.size() )
            {
                if( frm->i_type == T_AUTO || is_b( frm->i_type ) )
                    frm->i_type = T_P;
            }
            if( frm->i_type == T_BREF ) brefs++;
            if( frm->i_type == T_AUTO )
                frm->i_type = T_B;
            else if( !is_b( frm->i_type ) )
                break;
        }
        next[bframes]->i_bframes = bframes;
        if( p.b_pyramid && bframes > 1 && !brefs )
        {
            next[( bframes - 1 ) / 2]->i_type = T_BREF;
            brefs++;
        }
        // costs ahead of time for rate control (:1898-1935); VBV variants are out of scope
        if( !p.rc_is_cqp )
        {
            LaFrame *frames[BMAX + 3];
            int p1 = bframes + 1, b = bframes + 1, p0;
            frames[0] = last_nonb;
            for( int i = 0; i <= bframes; i++ ) frames[i + 1] = next[i];
            p0 = is_i( next[bframes]->i_type ) ? bframes + 1 : 0;
            frame_cost( frames, p0, p1, b );
            frames[b]->own_d0 = b - p0; frames[b]->own_d1 = 0;
            {
                const bool vbv_rows = ( p0 != p1 || bframes ) && p.vbv;
                if( vbv_rows )
                    frame_cost( frames, b, b, b ); // intra costs for the row sums (:1918-1919; memoized when already there)
                // the cell every B-frame of the mini-GOP is coded with (:1922-1933); VBV needs their row sums now
                int q0 = 0;
                for( int i = 1; i <= bframes; i++ )
                {
                    int q1 = bframes + 1;
                    if( frames[i]->i_type == T_B )
                        for( q1 = i; frames[q1]->i_type == T_B; ) q1++;
                    frames[i]->own_d0 = i - q0; frames[i]->own_d1 = q1 - i;
                    if( vbv_rows )
                        frame_cost( frames, q0, q1, i );
                    if( frames[i]->i_type == T_BREF ) q0 = i;
                }
            }
        }
        // The main-encode weight analysis of a P frame (:1937-1943, b_lookahead = 0) works on the full-resolution planes and
        // stays with the encoder, but its first step is visible in the lookahead's own outputs: when the luma statistics
        // call for a weight test it computes the frame's lowres intra costs if they are still missing (:365-370), which
        // fills i_cost_est[0][0] / i_cost_est_aq[0][0] of P frames no analysis has looked at yet.
        if( p.weightp >= 1 && next[bframes]->i_type == T_P && last_nonb && !next[bframes]->intra_calculated )
        {
            x264hip_weight guess, cand;
            if( weight_candidate( next[bframes], last_nonb, guess, cand ) )
            {
                LaFrame *one[1] = { next[bframes] };
                frame_cost( one, 0, 0, 0 );
            }
        }
        // coded order (:1945-1960)
        if( bframes )
        {
            std::vector<LaFrame *> tmp( bframes + 1 );
            int idx_list[2] = { brefs + 1, 1 };
            for( int i = 0; i < bframes; i++ )
            {
                int idx = idx_list[next[i]->i_type == T_BREF]++;
                tmp[idx] = next[i];
            }
            tmp[0] = next[bframes];
            for( int i = 0; i <= bframes; i++ ) next[i] = tmp[i];
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
        const int chunk = 64;
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
            for( LaFrame *f : res )
            {
                if( f->weights_prefetched || f == last_nonb ) continue;
                f->weights_prefetched = true;
                for( LaFrame *r : res )
                {
                    const int d = f->i_frame - r->i_frame;
                    if( d < 1 || d > p.dev.bframes + 1 ) continue;
                    x264hip_weight guess, cand;
                    if( !weight_candidate( f, r, guess, cand ) ) continue;
                    sf.push_back( f->slot ); sr.push_back( r->slot ); ws.push_back( cand );
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
    for( int i = n - 1; i >= 0; i-- ) la->L.free_slots.push_back( i );
    *out = la;
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
    x264hip_backend be = { ctx, dev_frame_put, dev_frame_stats, dev_weight_cost, dev_frame_cost, dev_prefetch, dev_mbtree, dev_qp_offsets, dev_put_batch, dev_prefetch_weights, dev_recalc, dev_row_satds, dev_frame_put_yuv, dev_add_quant_offsets, dev_put_batch_yuv };
    rc = x264hip_lookahead_open_backend( out, &p, &be );
    if( rc ) { x264hip_close( ctx ); return rc; }
    ( *out )->L.ctx = ctx;
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
    f->slot = L.free_slots.back(); L.free_slots.pop_back();
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
