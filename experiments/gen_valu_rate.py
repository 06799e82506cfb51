#!/usr/bin/env python3
"""Writes experiments/valu_rate_gen.hip: issue-rate microbenchmark of the instruction kinds the search kernel is made of.
One wave per workgroup, W waves per SIMD; every wave issues ITERS x 8 instructions of one kind on four independent registers.
Run on the GPU box: python experiments/gen_valu_rate.py && hipcc --offload-arch=gfx950 -O3 experiments/valu_rate_gen.hip -o /tmp/vr && /tmp/vr"""
OPS = [
    ("v_add_u32", "v_add_u32 {d}, {d}, {x}"),
    ("v_sub_u32", "v_sub_u32 {d}, {d}, {x}"),
    ("v_fma_f32", "v_fma_f32 {d}, {d}, {x}, {y}"),
    ("v_mov_b32", "v_mov_b32 {d}, {x}"),
    ("v_and_b32", "v_and_b32 {d}, {d}, {x}"),
    ("v_or_b32", "v_or_b32 {d}, {d}, {x}"),
    ("v_lshlrev_b32", "v_lshlrev_b32 {d}, 1, {d}"),
    ("v_ashrrev_i32", "v_ashrrev_i32 {d}, 1, {d}"),
    ("v_min_i32", "v_min_i32 {d}, {d}, {x}"),
    ("v_max_i32", "v_max_i32 {d}, {d}, {x}"),
    ("v_add3_u32", "v_add3_u32 {d}, {d}, {x}, {y}"),
    ("v_lshl_add_u32", "v_lshl_add_u32 {d}, {d}, 1, {x}"),
    ("v_lshl_or_b32", "v_lshl_or_b32 {d}, {d}, 1, {x}"),
    ("v_add_lshl_u32", "v_add_lshl_u32 {d}, {d}, {x}, 1"),
    ("v_bfe_u32", "v_bfe_u32 {d}, {d}, 1, 8"),
    ("v_med3_i32", "v_med3_i32 {d}, {d}, {x}, {y}"),
    ("v_cndmask_b32 vcc", "v_cndmask_b32 {d}, {d}, {x}, vcc"),
    ("v_cndmask_b32 sgpr pair", "v_cndmask_b32_e64 {d}, {d}, {x}, s[20:21]"),
    ("v_cmp_lt_i32 vcc", "v_cmp_lt_i32 vcc, {d}, {x}"),
    ("v_cmp_lt_i32 sgpr pair", "v_cmp_lt_i32_e64 s[20:21], {d}, {x}"),
    ("v_sad_u8", "v_sad_u8 {d}, {x}, {y}, {d}"),
    ("v_lerp_u8", "v_lerp_u8 {d}, {d}, {x}, {y}"),
    ("v_perm_b32", "v_perm_b32 {d}, {d}, {x}, {y}"),
    ("v_mad_i32_i24", "v_mad_i32_i24 {d}, {d}, {x}, {y}"),
    ("v_mul_u32_u24", "v_mul_u32_u24 {d}, {d}, {x}"),
    ("v_mul_lo_u32", "v_mul_lo_u32 {d}, {d}, {x}"),
    ("v_pk_sub_i16", "v_pk_sub_i16 {d}, {d}, {x}"),
    ("v_pk_add_u16", "v_pk_add_u16 {d}, {d}, {x}"),
    ("v_pk_max_i16", "v_pk_max_i16 {d}, {d}, {x}"),
    ("v_pk_mad_i16", "v_pk_mad_i16 {d}, {d}, {x}, {y}"),
    ("v_dot2_u32_u16", "v_dot2_u32_u16 {d}, {x}, {y}, {d}"),
    ("v_add_u32_dpp quad_perm", "v_add_u32_dpp {d}, {d}, {d} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    ("v_add_u32_dpp row_ror", "v_add_u32_dpp {d}, {d}, {d} row_ror:4 row_mask:0xf bank_mask:0xf"),
    ("v_mov_b32_dpp row_ror", "v_mov_b32_dpp {d}, {d} row_ror:8 row_mask:0xf bank_mask:0xf"),
    ("v_min_i32_sdwa", "v_min_i32_sdwa {d}, {d}, {x} dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"),
    ("v_readfirstlane_b32", "v_readfirstlane_b32 s22, {d}"),
    ("s_add_u32", "s_add_u32 s22, s22, 1"),
    ("s_nop 0", "s_nop 0"),
    ("s_nop 1", "s_nop 1"),
    ("v_cndmask_b32 vcc, dst != src", "v_cndmask_b32 {d}, {x}, {y}, vcc"),
    ("v_cndmask_b32_e64 vcc", "v_cndmask_b32_e64 {d}, {d}, {x}, vcc"),
    ("v_addc_co_u32 vcc", "v_addc_co_u32 {d}, vcc, {d}, {x}, vcc"),
    ("v_cmp vcc + v_cndmask vcc pairs", "PAIR:v_cmp_lt_i32 vcc, {d}, {x}|v_cndmask_b32 {d}, {d}, {y}, vcc"),
    ("v_cmp sgpr + v_cndmask sgpr pairs", "PAIR:v_cmp_lt_i32_e64 s[20:21], {d}, {x}|v_cndmask_b32_e64 {d}, {d}, {y}, s[20:21]"),
    ("v_cmp vcc + s_nop 1 + v_cndmask vcc", "TRIPLE:v_cmp_lt_i32 vcc, {d}, {x}|s_nop 1|v_cndmask_b32 {d}, {d}, {y}, vcc"),
    ("v_cmp vcc + 3 v_cndmask vcc (x2)", "RAW:v_cmp_lt_i32 vcc, %0, %4|v_cndmask_b32 %0, %0, %5, vcc|v_cndmask_b32 %1, %1, %5, vcc|v_cndmask_b32 %2, %2, %5, vcc|v_cmp_lt_i32 vcc, %3, %4|v_cndmask_b32 %3, %3, %5, vcc|v_cndmask_b32 %1, %1, %4, vcc|v_cndmask_b32 %2, %2, %4, vcc"),
    ("v_cmp sgpr + 3 v_cndmask sgpr (x2)", "RAW:v_cmp_lt_i32_e64 s[20:21], %0, %4|v_cndmask_b32_e64 %0, %0, %5, s[20:21]|v_cndmask_b32_e64 %1, %1, %5, s[20:21]|v_cndmask_b32_e64 %2, %2, %5, s[20:21]|v_cmp_lt_i32_e64 s[20:21], %3, %4|v_cndmask_b32_e64 %3, %3, %5, s[20:21]|v_cndmask_b32_e64 %1, %1, %4, s[20:21]|v_cndmask_b32_e64 %2, %2, %4, s[20:21]"),
    ("v_cndmask vcc x8 after one v_cmp", "RAW:v_cmp_lt_i32 vcc, %0, %4|v_cndmask_b32 %0, %0, %5, vcc|v_cndmask_b32 %1, %1, %5, vcc|v_cndmask_b32 %2, %2, %5, vcc|v_cndmask_b32 %3, %3, %4, vcc|v_cndmask_b32 %0, %0, %4, vcc|v_cndmask_b32 %1, %1, %4, vcc|v_cndmask_b32 %2, %2, %4, vcc"),
    ("cmp, add, cndmask e32 | x2 + 2 add", "RAW:v_cmp_lt_i32 vcc, %0, %4|v_add_u32 %1, %1, %4|v_cndmask_b32 %0, %0, %5, vcc|v_add_u32 %2, %2, %4|v_cmp_lt_i32 vcc, %3, %4|v_add_u32 %1, %1, %4|v_cndmask_b32 %3, %3, %5, vcc|v_add_u32 %2, %2, %4"),
    ("cmp, cndmask e32, add, cndmask e32 | x2", "RAW:v_cmp_lt_i32 vcc, %0, %4|v_cndmask_b32 %0, %0, %5, vcc|v_add_u32 %1, %1, %4|v_cndmask_b32 %2, %2, %5, vcc|v_cmp_lt_i32 vcc, %3, %4|v_cndmask_b32 %3, %3, %5, vcc|v_add_u32 %1, %1, %4|v_cndmask_b32 %2, %2, %4, vcc"),
    ("add / cndmask e32 alternating, no cmp", "RAW:v_add_u32 %0, %0, %4|v_cndmask_b32 %1, %1, %5, vcc|v_add_u32 %2, %2, %4|v_cndmask_b32 %3, %3, %5, vcc|v_add_u32 %0, %0, %4|v_cndmask_b32 %1, %1, %5, vcc|v_add_u32 %2, %2, %4|v_cndmask_b32 %3, %3, %5, vcc"),
    ("cndmask e32 x2 then 2 adds | x2", "RAW:v_cndmask_b32 %0, %0, %5, vcc|v_cndmask_b32 %1, %1, %5, vcc|v_add_u32 %2, %2, %4|v_add_u32 %3, %3, %4|v_cndmask_b32 %0, %0, %5, vcc|v_cndmask_b32 %1, %1, %5, vcc|v_add_u32 %2, %2, %4|v_add_u32 %3, %3, %4"),
    ("cndmask e32 dst chain distinct x8 (4 regs src fixed)", "RAW:v_cndmask_b32 %0, %4, %5, vcc|v_cndmask_b32 %1, %4, %5, vcc|v_cndmask_b32 %2, %4, %5, vcc|v_cndmask_b32 %3, %4, %5, vcc|v_cndmask_b32 %0, %5, %4, vcc|v_cndmask_b32 %1, %5, %4, vcc|v_cndmask_b32 %2, %5, %4, vcc|v_cndmask_b32 %3, %5, %4, vcc"),
    ("v_add_u32 / s_add_u32 alternating", None),
    ("v_add_u32 / v_sad_u8 alternating", None),
]
REGS = ["%0", "%1", "%2", "%3"]
def body(i, name, t):
    if t is None:
        other = "s_add_u32 s22, s22, 1" if "s_add" in name else "v_sad_u8 {d}, %4, %5, {d}"
        lines = []
        for k in range(4):
            lines.append("v_add_u32 {d}, {d}, %4".format(d=REGS[k]))
            lines.append(other.format(d=REGS[(k + 2) % 4]))
    elif t.startswith("RAW:"):
        lines = t[4:].split("|")
    elif t.startswith("PAIR:") or t.startswith("TRIPLE:"):
        parts = t.split(":", 1)[1].split("|")
        lines = [q.format(d=REGS[k], x="%4", y="%5") for k in range(4) for q in parts]
    else:
        lines = [t.format(d=REGS[k % 4], x="%4", y="%5") for k in range(8)]
    return '        if( KIND == %d ) asm volatile( "%s" : "+v"( a ), "+v"( b ), "+v"( c ), "+v"( d ) : "v"( x ), "v"( y ) : "vcc", "scc", "s20", "s21", "s22" );' % (i, "\\n".join(lines))
src = r'''// generated by experiments/gen_valu_rate.py -- do not edit
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 20000
template <int KIND>
__global__ __launch_bounds__( 64 ) void rate_kernel( uint32_t *out, uint32_t x, uint32_t y )
{
    uint32_t a = threadIdx.x, b = x, c = y, d = x ^ y;
    asm volatile( "s_mov_b64 s[20:21], exec\ns_mov_b64 vcc, exec\ns_mov_b32 s22, 0" : : : "vcc", "s20", "s21", "s22" );
    for( int i = 0; i < ITERS; i++ )
    {
%s
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
static const char *names[] = { %s };
template <int KIND>
static void run( uint32_t *out, int waves_per_simd )
{
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate( &e0 ); hipEventCreate( &e1 );
    rate_kernel<KIND><<<grid, 64>>>( out, 3, 5 );
    hipEventRecord( e0 );
    rate_kernel<KIND><<<grid, 64>>>( out, 3, 5 );
    hipEventRecord( e1 );
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime( &ms, e0, e1 );
    printf( "%%-36s waves/SIMD %%d: %%.3f ns per instruction per SIMD\n", names[KIND], waves_per_simd, ms * 1e6 / ( 8.0 * ITERS * waves_per_simd ) ); // PAIR/TRIPLE kinds: per 8 instructions' worth = per pair x 4 / 8
}
int main()
{
    uint32_t *out;
    hipMalloc( &out, 256 * 4 * 8 * 64 * 4 );
    const int ws[3] = { 1, 4, 8 };
    for( int wi = 0; wi < 3; wi++ )
    {
%s
    }
    return 0;
}
''' % ("\n".join(body(i, n, t) for i, (n, t) in enumerate(OPS)), ", ".join('"%s"' % n for n, _ in OPS),
       "\n".join("        run<%d>( out, ws[wi] );" % i for i in range(len(OPS))))
open(__file__.replace("gen_valu_rate.py", "valu_rate_gen.hip"), "w").write(src)
