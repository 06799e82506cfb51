// TEST TOOL: compiles the arithmetic of x264_amd/csrc/block_metrics.h (the functions the device kernels call) for the host, so
// that tests/test_block_metrics_host.py can check it against the oracle without a GPU.  Not part of libx264hip.so.
#define BM_HD inline
#include "../../x264_amd/csrc/block_metrics.h"
#include "../../x264_amd/csrc/dct_quant_block.h"

template <typename T>
static uint64_t run( int metric, int w, int h, const T *a, long sa, const T *b, long sb )
{
#define CASE( M, W, H ) if( metric == M && w == W && h == H ) return bm_block<T, M, W, H>( a, sa, b, sb );
    CASE( BM_SSD, 16, 16 ) CASE( BM_SSD, 16, 8 ) CASE( BM_SSD, 8, 16 ) CASE( BM_SSD, 8, 8 ) CASE( BM_SSD, 8, 4 ) CASE( BM_SSD, 4, 8 ) CASE( BM_SSD, 4, 4 )
    CASE( BM_SA8D, 16, 16 ) CASE( BM_SA8D, 8, 8 )
    CASE( BM_VAR, 16, 16 ) CASE( BM_VAR, 8, 16 ) CASE( BM_VAR, 8, 8 )
    CASE( BM_HADAMARD_AC, 16, 16 ) CASE( BM_HADAMARD_AC, 16, 8 ) CASE( BM_HADAMARD_AC, 8, 16 ) CASE( BM_HADAMARD_AC, 8, 8 )
    CASE( BM_VSAD, 16, 16 ) CASE( BM_VSAD, 16, 8 )
    CASE( BM_ASD8, 8, 16 ) CASE( BM_ASD8, 8, 8 )
#undef CASE
    return ~0ull;
}
extern "C" uint64_t bm_host_u8( int metric, int w, int h, const uint8_t *a, long sa, const uint8_t *b, long sb ) { return run<uint8_t>( metric, w, h, a, sa, b, sb ); }
extern "C" uint64_t bm_host_u16( int metric, int w, int h, const uint16_t *a, long sa, const uint16_t *b, long sb ) { return run<uint16_t>( metric, w, h, a, sa, b, sb ); }

extern "C" int dq8x8_host_u8( const uint8_t *fe, long fs, const uint8_t *fd, long ds, const uint32_t *mf, const uint32_t *bias, int16_t *out ) { return dq_block8x8<uint8_t, int16_t>( fe, fs, fd, ds, mf, bias, out ); }
extern "C" int dq8x8_host_u16( const uint16_t *fe, long fs, const uint16_t *fd, long ds, const uint32_t *mf, const uint32_t *bias, int32_t *out ) { return dq_block8x8<uint16_t, int32_t>( fe, fs, fd, ds, mf, bias, out ); }

// the main-encode motion search of x264_amd/csrc/me_full.h (body of me_full_kernel) on host memory
#include <stdlib.h>
#include "../../x264_amd/csrc/me_full.h"
struct MfHostReq  // C layout handed over by the Python test (pointers as given)
{
    int i_pixel, me_method, subpel_refine, me_range, mbcmp_satd, fpelcmp_satd;
    const void *fenc; int fenc_stride;
    const void *ref[4]; int stride;
    const uint16_t *integral; long integral_lower;
    int mvp[2], lim_min[2], lim_max[2], spel_min[2], spel_max[2];
    const uint16_t *cost_mv;
};
template <typename T>
static void mf_run( const MfHostReq *h, const int16_t *mvc, int n_mvc, int *out )
{
    MfReq<T> r;
    r.i_pixel = h->i_pixel; r.me_method = h->me_method; r.subpel_refine = h->subpel_refine; r.me_range = h->me_range;
    r.mbcmp_satd = h->mbcmp_satd; r.fpelcmp_satd = h->fpelcmp_satd;
    r.fenc = (const T *)h->fenc; r.fenc_stride = h->fenc_stride;
    for( int k = 0; k < 4; k++ ) r.ref[k] = (const T *)h->ref[k];
    r.stride = h->stride; r.integral = h->integral; r.integral_lower = h->integral_lower;
    for( int k = 0; k < 2; k++ ) { r.mvp[k] = h->mvp[k]; r.lim_min[k] = h->lim_min[k]; r.lim_max[k] = h->lim_max[k]; r.spel_min[k] = h->spel_min[k]; r.spel_max[k] = h->spel_max[k]; }
    r.cost_mv = h->cost_mv;
    r.scratch = malloc( (size_t)MF_TESA_ROWS_MAX * MF_TESA_WIDTH_MAX * 12 );
    mefull::mf_me_search_full<T>( &r, (const int16_t( * )[2])mvc, n_mvc, out );
    free( r.scratch );
}
extern "C" void mf_host_u8( const MfHostReq *h, const int16_t *mvc, int n_mvc, int *out ) { mf_run<uint8_t>( h, mvc, n_mvc, out ); }
extern "C" void mf_host_u16( const MfHostReq *h, const int16_t *mvc, int n_mvc, int *out ) { mf_run<uint16_t>( h, mvc, n_mvc, out ); }

// The cooperative form of the same search on the host: W threads stand for the lanes of a wave.  Every thread runs the whole request,
// the collectives of the Coop interface (ballot, packed minimum, prefix minimum, broadcast) go through a shared slot array between two
// barriers -- so the parts of me_full.h that are written for several lanes (the chunked scans, the ordered compaction of the ads
// survivors, the SAD stage's thresholds as a prefix minimum, the survivor thinning) run without a GPU, against the one-thread form.
#include <pthread.h>
#include <thread>
#include <vector>
struct CoopShared
{
    pthread_barrier_t bar;
    unsigned long long slot[64];
    int16_t xs[MF_TESA_WIDTH_MAX + 64];
};
template <int LANES>
struct CoopThreads
{
    static constexpr int W = LANES;
    CoopShared *sh;
    int l;
    int lane() const { return l; }
    void sync() const { pthread_barrier_wait( &sh->bar ); }
    // every lane's value, as lane k's view of slot[k]
    void share( unsigned long long v, unsigned long long *all ) const
    {
        sh->slot[l] = v;
        sync();
        for( int k = 0; k < W; k++ ) all[k] = sh->slot[k];
        sync();
    }
    void argmin( int &cost, int &idx ) const
    {
        unsigned long long a[W], key;
        share( ( (unsigned long long)(unsigned)cost << 32 ) | (unsigned)idx, a );
        key = a[0];
        for( int k = 1; k < W; k++ ) key = a[k] < key ? a[k] : key;
        cost = (int)( key >> 32 ); idx = (int)(unsigned)key;
    }
    unsigned long long ballot( bool p ) const
    {
        unsigned long long a[W], m = 0;
        share( p ? 1ull : 0ull, a );
        for( int k = 0; k < W; k++ ) m |= a[k] << k;
        return m;
    }
    unsigned long long max64( unsigned long long v ) const
    {
        unsigned long long a[W];
        share( v, a );
        for( int k = 0; k < W; k++ ) v = a[k] > v ? a[k] : v;
        return v;
    }
    int bcast( int v, int k ) const
    {
        unsigned long long a[W];
        share( (unsigned long long)(unsigned)v, a );
        return (int)(unsigned)a[k];
    }
    void min_scan( int v, int &before, int &all ) const
    {
        unsigned long long a[W];
        share( (unsigned long long)(unsigned)v, a );
        before = 1 << 28; all = 1 << 28;
        for( int k = 0; k < W; k++ )
        {
            const int x = (int)(unsigned)a[k];
            if( k < l && x < before ) before = x;
            if( x < all ) all = x;
        }
    }
    int16_t *xs() { return sh->xs; }
};
template <typename T, int LANES>
static int mf_run_lanes( const MfHostReq *h, const int16_t *mvc, int n_mvc, int *out )
{
    MfReq<T> r;
    r.i_pixel = h->i_pixel; r.me_method = h->me_method; r.subpel_refine = h->subpel_refine; r.me_range = h->me_range;
    r.mbcmp_satd = h->mbcmp_satd; r.fpelcmp_satd = h->fpelcmp_satd;
    r.fenc = (const T *)h->fenc; r.fenc_stride = h->fenc_stride;
    for( int k = 0; k < 4; k++ ) r.ref[k] = (const T *)h->ref[k];
    r.stride = h->stride; r.integral = h->integral; r.integral_lower = h->integral_lower;
    for( int k = 0; k < 2; k++ ) { r.mvp[k] = h->mvp[k]; r.lim_min[k] = h->lim_min[k]; r.lim_max[k] = h->lim_max[k]; r.spel_min[k] = h->spel_min[k]; r.spel_max[k] = h->spel_max[k]; }
    r.cost_mv = h->cost_mv;
    r.scratch = malloc( (size_t)MF_TESA_ROWS_MAX * MF_TESA_WIDTH_MAX * 12 );
    CoopShared sh;
    pthread_barrier_init( &sh.bar, nullptr, LANES );
    int res[LANES][4];
    std::vector<std::thread> th;
    for( int l = 0; l < LANES; l++ )
        th.emplace_back( [&, l]() {
            CoopThreads<LANES> coop;
            coop.sh = &sh; coop.l = l;
            mefull::mf_me_search_full<T, CoopThreads<LANES>>( &r, (const int16_t( * )[2])mvc, n_mvc, res[l], coop );
        } );
    for( auto &t : th ) t.join();
    pthread_barrier_destroy( &sh.bar );
    free( r.scratch );
    int same = 1;
    for( int l = 1; l < LANES; l++ )
        for( int k = 0; k < 4; k++ )
            same &= res[l][k] == res[0][k]; // the lanes of a wave hold identical state at the end
    for( int k = 0; k < 4; k++ ) out[k] = res[0][k];
    return same;
}
extern "C" int mf_host_lanes_u8( const MfHostReq *h, const int16_t *mvc, int n_mvc, int *out ) { return mf_run_lanes<uint8_t, 8>( h, mvc, n_mvc, out ); }
extern "C" int mf_host_lanes_u16( const MfHostReq *h, const int16_t *mvc, int n_mvc, int *out ) { return mf_run_lanes<uint16_t, 8>( h, mvc, n_mvc, out ); }

// integral planes (x264_amd/csrc/integral.h): the two passes of the device kernels run as loops over host memory
#include "../../x264_amd/csrc/integral.h"
template <typename T>
static void ii_run( const T *plane, long stride, int width, int height, uint16_t *sum8, uint16_t *sum4 )
{
    uint16_t *h8 = (uint16_t *)calloc( (size_t)stride * height, 2 ), *h4 = (uint16_t *)calloc( (size_t)stride * height, 2 );
    for( int y = 0; y < height; y++ )
        for( int x = 0; x < width; x++ )
            ii_row_sums<T>( plane + (long)y * stride, x, width, h8 + (long)y * stride + x, h4 + (long)y * stride + x );
    for( int y = 0; y < height; y++ )
        for( int x = 0; x < width; x++ )
        {
            if( x + 8 <= width && y + 8 <= height ) sum8[(long)y * stride + x] = ii_col_sum( h8 + (long)y * stride + x, stride, 8 );
            if( x + 4 <= width && y + 4 <= height ) sum4[(long)y * stride + x] = ii_col_sum( h4 + (long)y * stride + x, stride, 4 );
        }
    free( h8 ); free( h4 );
}
extern "C" void ii_host_u8( const uint8_t *plane, long stride, int width, int height, uint16_t *sum8, uint16_t *sum4 ) { ii_run<uint8_t>( plane, stride, width, height, sum8, sum4 ); }
extern "C" void ii_host_u16( const uint16_t *plane, long stride, int width, int height, uint16_t *sum8, uint16_t *sum4 ) { ii_run<uint16_t>( plane, stride, width, height, sum8, sum4 ); }
