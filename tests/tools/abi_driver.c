/* TEST INFRASTRUCTURE: a plain C caller of the library, the way the reference encoder would bind it (INTEGRATION.md sections 1-3).
 * Includes include/x264hip.h, dlopen()s libx264hip.so, resolves the entry points by name and runs the call sequence of the
 * slicetype_frame_cost hook for a P evaluation and a B evaluation:
 *     x264hip_open -> x264hip_frame_put (host luma, device AQ) x3 -> x264hip_frame_cost( P: 0 <- 2, first trigger: searches L0 )
 *     -> x264hip_frame_cost( B: 0 <- 1 -> 2, searches both lists ) -> x264hip_get_mvs / x264hip_get_lowres_costs / x264hip_get_intra_costs
 * and compares every returned array with the expected files written by tests/test_abi_c_driver.py (from the oracle).
 *   usage: abi_driver <libx264hip.so> <dir>       (dir holds params.txt, frames.bin, cost_mv.bin and expect_*.bin)
 * Exit code 0 = every array identical. */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "x264hip.h"

static void *lib;
#define RESOLVE( name ) __typeof__( name ) *p_##name = (__typeof__( name ) *)dlsym( lib, #name ); \
    if( !p_##name ) { fprintf( stderr, "missing symbol %s\n", #name ); return 2; }

static void *read_file( const char *dir, const char *name, size_t bytes )
{
    char path[1024];
    snprintf( path, sizeof( path ), "%s/%s", dir, name );
    FILE *f = fopen( path, "rb" );
    if( !f ) { fprintf( stderr, "cannot open %s\n", path ); exit( 2 ); }
    void *buf = malloc( bytes ? bytes : 1 );
    if( fread( buf, 1, bytes, f ) != bytes ) { fprintf( stderr, "short read on %s\n", path ); exit( 2 ); }
    fclose( f );
    return buf;
}

static int same( const char *what, const void *got, const char *dir, const char *name, size_t bytes )
{
    void *want = read_file( dir, name, bytes );
    int ok = !memcmp( got, want, bytes );
    if( !ok )
    {
        size_t i = 0;
        while( ( (const unsigned char *)got )[i] == ( (const unsigned char *)want )[i] ) i++;
        fprintf( stderr, "MISMATCH %s: first difference at byte %zu of %zu\n", what, i, bytes );
    }
    free( want );
    return ok;
}

int main( int argc, char **argv )
{
    if( argc < 3 ) { fprintf( stderr, "usage: %s <libx264hip.so> <dir>\n", argv[0] ); return 2; }
    const char *dir = argv[2];
    lib = dlopen( argv[1], RTLD_NOW | RTLD_LOCAL );
    if( !lib ) { fprintf( stderr, "dlopen: %s\n", dlerror() ); return 2; }
    RESOLVE( x264hip_open ) RESOLVE( x264hip_close ) RESOLVE( x264hip_frame_put ) RESOLVE( x264hip_frame_cost ) RESOLVE( x264hip_get_mvs )
    RESOLVE( x264hip_get_lowres_costs ) RESOLVE( x264hip_get_intra_costs ) RESOLVE( x264hip_get_inv_qscale ) RESOLVE( x264hip_geometry )
    RESOLVE( x264hip_strerror ) RESOLVE( x264hip_frame_stats ) RESOLVE( x264hip_synchronize )

    /* params.txt: width height bit_depth bframes lambda me_method subpel_refine me_range mv_range subme mbcmp_satd fpelcmp_satd */
    char path[1024];
    snprintf( path, sizeof( path ), "%s/params.txt", dir );
    FILE *f = fopen( path, "r" );
    if( !f ) { fprintf( stderr, "cannot open %s\n", path ); return 2; }
    x264hip_params p;
    memset( &p, 0, sizeof( p ) );
    if( fscanf( f, "%d %d %d %d %d %d %d %d %d %d %d %d", &p.width, &p.height, &p.bit_depth, &p.bframes, &p.lambda, &p.me_method, &p.subpel_refine, &p.me_range,
                &p.mv_range, &p.subme, &p.mbcmp_satd, &p.fpelcmp_satd ) != 12 ) { fprintf( stderr, "bad params.txt\n" ); return 2; }
    fclose( f );
    p.weighted_bipred = 1; p.aq_mode = 1; p.aq_strength = 1.0f; p.max_frames = 4;
    const int n_tab = 2 * 4 * p.mv_range;
    uint16_t *cost_mv = read_file( dir, "cost_mv.bin", ( 2 * (size_t)n_tab + 1 ) * 2 );
    p.cost_mv = cost_mv + n_tab; /* centred, like h->cost_mv[X264_LOOKAHEAD_QP] */
    const int psz = p.bit_depth == 8 ? 1 : 2;
    const size_t frame_b = (size_t)p.width * p.height * psz;
    unsigned char *frames = read_file( dir, "frames.bin", 3 * frame_b );

    x264hip_ctx *ctx = NULL;
    int rc = p_x264hip_open( &ctx, 0, &p );
    if( rc ) { fprintf( stderr, "x264hip_open: %s\n", p_x264hip_strerror( rc ) ); return 3; }
    int mb_w, mb_h, stride;
    p_x264hip_geometry( ctx, &mb_w, &mb_h, &stride );
    const int n_mb = mb_w * mb_h;
#define CK( call ) do { rc = ( call ); if( rc ) { fprintf( stderr, "%s -> %s\n", #call, p_x264hip_strerror( rc ) ); return 3; } } while( 0 )
    for( int i = 0; i < 3; i++ ) /* x264_frame_init_lowres + x264_adaptive_quant_frame of every input frame */
        CK( p_x264hip_frame_put( ctx, i, frames + i * frame_b, p.width, 0, NULL, NULL, 0, NULL ) );

    int ok = 1;
    int16_t *mvs = malloc( (size_t)n_mb * 4 );
    int *mv_costs = malloc( (size_t)n_mb * 4 ), *rows = malloc( (size_t)mb_h * 4 );
    uint16_t *lc = malloc( (size_t)n_mb * 2 );
    uint64_t sum, ssd, stats[2];
    CK( p_x264hip_frame_stats( ctx, 0, &sum, &ssd ) );
    stats[0] = sum; stats[1] = ssd;
    ok &= same( "i_pixel_sum / i_pixel_ssd of frame 0", stats, dir, "expect_stats0.bin", 16 );
    CK( p_x264hip_get_inv_qscale( ctx, 0, lc ) );
    ok &= same( "i_inv_qscale_factor of frame 0", lc, dir, "expect_invq0.bin", (size_t)n_mb * 2 );

    /* slicetype_frame_cost( p0 = 0, p1 = 2, b = 2 ): first trigger of lowres_mvs[0][1], no weight, intra not yet calculated */
    x264hip_cost c;
    const int search_p[2] = { 1, 0 }, search_b[2] = { 1, 1 };
    CK( p_x264hip_frame_cost( ctx, 0, 2, 2, 2, 0, search_p, NULL, 1, 0, &c ) );
    int sums[5] = { c.cost_est, c.cost_est_aq, c.intra_mbs, c.intra_cost_est, c.intra_cost_est_aq };
    ok &= same( "P evaluation sums", sums, dir, "expect_p_sums.bin", sizeof( sums ) );
    CK( p_x264hip_get_mvs( ctx, 2, 0, 1, mvs, mv_costs ) );
    ok &= same( "lowres_mvs[0][1] of frame 2", mvs, dir, "expect_p_mvs.bin", (size_t)n_mb * 4 );
    ok &= same( "lowres_mv_costs[0][1] of frame 2", mv_costs, dir, "expect_p_mvcosts.bin", (size_t)n_mb * 4 );
    CK( p_x264hip_get_lowres_costs( ctx, 2, 2, 0, lc, rows ) );
    ok &= same( "lowres_costs[2][0] of frame 2", lc, dir, "expect_p_lc.bin", (size_t)n_mb * 2 );
    ok &= same( "i_row_satds[2][0] of frame 2", rows, dir, "expect_p_rows.bin", (size_t)mb_h * 4 );
    CK( p_x264hip_get_intra_costs( ctx, 2, lc ) );
    ok &= same( "i_intra_cost of frame 2", lc, dir, "expect_p_intra.bin", (size_t)n_mb * 2 );

    /* slicetype_frame_cost( 0, 2, 1 ): both searches of frame 1, the list-1 reference's own L0 field exists (slicetype.c:629) */
    CK( p_x264hip_frame_cost( ctx, 0, 2, 1, 1, 1, search_b, NULL, 1, 1, &c ) );
    int sums_b[2] = { c.cost_est, c.cost_est_aq };
    ok &= same( "B evaluation sums", sums_b, dir, "expect_b_sums.bin", sizeof( sums_b ) );
    CK( p_x264hip_get_lowres_costs( ctx, 1, 1, 1, lc, rows ) );
    ok &= same( "lowres_costs[1][1] of frame 1", lc, dir, "expect_b_lc.bin", (size_t)n_mb * 2 );
    ok &= same( "i_row_satds[1][1] of frame 1", rows, dir, "expect_b_rows.bin", (size_t)mb_h * 4 );
    CK( p_x264hip_synchronize( ctx ) );
    p_x264hip_close( ctx );
    dlclose( lib );
    printf( ok ? "abi_driver: all arrays identical\n" : "abi_driver: MISMATCH\n" );
    return ok ? 0 : 1;
}
