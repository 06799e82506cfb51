/* shard_driver.c -- the window shard of include/x264hip.h from plain C, the way INTEGRATION.md section 7 shows it: the library through
 * dlopen (like common/opencl.c:53-61 loads its runtime), one rank over the RCCL transport in loop-back mode, device memory for the
 * pictures through the HIP runtime's C entry points (also resolved with dlsym: this program links nothing but libdl).
 * usage: shard_driver <libx264hip.so> <dir with frames.bin params.bin cost_mv.bin> <W> <H> <n_frames> <sizeof params blob>
 * prints "frame <display number> <type> <i_cost_est[0][0]>" per frame in coded order (tests/test_gpu_c_shard.py compares). */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "x264hip.h"

#define RESOLVE( name ) __typeof__( name ) *p_##name = (__typeof__( name ) *)dlsym( lib, #name ); if( !p_##name ) { fprintf( stderr, "missing %s\n", #name ); return 2; }

static void *slurp( const char *dir, const char *name, size_t *n )
{
    char path[1024];
    snprintf( path, sizeof( path ), "%s/%s", dir, name );
    FILE *f = fopen( path, "rb" );
    if( !f ) { perror( path ); exit( 2 ); }
    fseek( f, 0, SEEK_END ); *n = (size_t)ftell( f ); fseek( f, 0, SEEK_SET );
    void *p = malloc( *n ? *n : 1 );
    if( fread( p, 1, *n, f ) != *n ) { perror( path ); exit( 2 ); }
    fclose( f );
    return p;
}

int main( int argc, char **argv )
{
    if( argc < 7 ) return 2;
    void *lib = dlopen( argv[1], RTLD_NOW );
    if( !lib ) { fprintf( stderr, "%s\n", dlerror() ); return 2; }
    RESOLVE( x264hip_rccl_unique_id ) RESOLVE( x264hip_shard_transport_rccl ) RESOLVE( x264hip_shard_open ) RESOLVE( x264hip_shard_lookahead )
    RESOLVE( x264hip_shard_put_frames ) RESOLVE( x264hip_shard_status ) RESOLVE( x264hip_shard_stats ) RESOLVE( x264hip_shard_loopback_verify )
    RESOLVE( x264hip_shard_close ) RESOLVE( x264hip_lookahead_get_frame ) RESOLVE( x264hip_strerror )
    void *hip = dlopen( "libamdhip64.so", RTLD_NOW );
    if( !hip ) { fprintf( stderr, "%s\n", dlerror() ); return 2; }
    int ( *hip_malloc )( void **, size_t ) = (int ( * )( void **, size_t ))dlsym( hip, "hipMalloc" );
    int ( *hip_memcpy )( void *, const void *, size_t, int ) = (int ( * )( void *, const void *, size_t, int ))dlsym( hip, "hipMemcpy" );
    int ( *hip_free )( void * ) = (int ( * )( void * ))dlsym( hip, "hipFree" );
    if( !hip_malloc || !hip_memcpy || !hip_free ) return 2;

    const char *dir = argv[2];
    const int W = atoi( argv[3] ), H = atoi( argv[4] ), nf = atoi( argv[5] );
    size_t n_frames_b, n_params, n_tab;
    unsigned char *frames = slurp( dir, "frames.bin", &n_frames_b );
    x264hip_la_params *params = slurp( dir, "params.bin", &n_params );
    uint16_t *tab = slurp( dir, "cost_mv.bin", &n_tab );
    if( n_params != sizeof( x264hip_la_params ) || (size_t)atoi( argv[6] ) != n_params || n_frames_b != (size_t)W * H * nf )
    {
        fprintf( stderr, "params blob is %zu bytes, x264hip_la_params %zu\n", n_params, sizeof( x264hip_la_params ) );
        return 2;
    }
    params->dev.cost_mv = tab + ( n_tab / 2 - 1 ) / 2; /* the centred table (h->cost_mv[qp], analyse.c:151-157) */

#define CK( call ) do { int rc_ = ( call ); if( rc_ ) { fprintf( stderr, "%s -> %s\n", #call, p_x264hip_strerror( rc_ ) ); return 3; } } while( 0 )
    unsigned char id[128];
    x264hip_shard_transport t;
    CK( p_x264hip_rccl_unique_id( id ) );
    CK( p_x264hip_shard_transport_rccl( &t, id, 0, 1, 0 ) );
    t.loopback = 1;
    x264hip_shard *sh = NULL;
    CK( p_x264hip_shard_open( &sh, 0, params, &t ) );
    void *clip = NULL;
    if( hip_malloc( &clip, n_frames_b ) || hip_memcpy( clip, frames, n_frames_b, 1 /* hipMemcpyHostToDevice */ ) ) { fprintf( stderr, "device clip\n" ); return 3; }
    const void **ptrs = malloc( sizeof( void * ) * nf );
    for( int i = 0; i < nf; i++ ) ptrs[i] = (unsigned char *)clip + (size_t)i * W * H;
    CK( p_x264hip_shard_put_frames( sh, 0, nf, ptrs, W ) );
    x264hip_lookahead *la = p_x264hip_shard_lookahead( sh );
    for( int got = 1; got; )
    {
        x264hip_la_frame out;
        CK( p_x264hip_lookahead_get_frame( la, 1, &out, &got ) );
        if( got ) printf( "frame %d %d %d\n", out.frame, out.type, out.cost_est[0][0] );
    }
    CK( p_x264hip_shard_status( sh ) );
    int checked = 0;
    CK( p_x264hip_shard_loopback_verify( sh, &checked ) );
    uint64_t st[X264HIP_SHARD_STATS];
    CK( p_x264hip_shard_stats( sh, st, X264HIP_SHARD_STATS ) );
    printf( "loopback checks %d, chunks %llu, fields %llu, cells %llu\n", checked, (unsigned long long)st[X264HIP_SHARD_CHUNKS],
            (unsigned long long)st[X264HIP_SHARD_FIELDS_SEARCHED], (unsigned long long)st[X264HIP_SHARD_CELLS_EVALUATED] );
    p_x264hip_shard_close( sh );
    hip_free( clip );
    return 0;
}
