// strip_layout.h -- index arithmetic of the strip copy of a padded plane (host and device; ME_HD comes from the includer).
//
// A padded plane is `rows` rows of `stride` samples (stride a multiple of 8), origin of the picture at (LA_PAD, LA_PAD).  Its strip
// copy cuts it into stride/8 vertical strips: strip k holds columns 8k .. 8k+15 of EVERY row -- 16 samples per row, the rows of a
// strip one after the other -- so every column is stored twice (as the right half of strip k-1 and the left half of strip k) and
// a plane's strips take twice the plane.  Consequences the search kernel is built on (me_search.h):
//   * the 8 rows of an 8x8 block candidate are 8 x 16 samples apart = 128 consecutive samples of memory (2-3 cache lines of
//     8-bit samples) whatever the candidate's position;
//   * 8 samples starting at ANY column c live inside strip c >> 3 at offset c & 7 (0..7, so the read ends at offset <= 14), and so
//     do the 8 samples starting one column further right (offset <= 8 + 7 = 15): the two taps of a quarter-pel sample never
//     straddle strips.
// The last strip of a plane has no right half (those columns do not exist); no legal candidate reads it.
#pragma once
namespace strip_layout
{
// samples of one strip (all rows)
ME_HD int strip_elems( int rows ) { return rows * 16; }
// samples of one plane's strips
ME_HD long plane_elems( int rows, int stride ) { return (long)( stride >> 3 ) * rows * 16; }
// writer side: offset of sample 0 of row Y of strip k inside a plane's strips
ME_HD long row_off( int k, int Y, int rows ) { return ( (long)k * rows + Y ) * 16; }
// reader side: offset of the 8 samples that start at padded column c of the row whose strip-row offset (16 * padded row) is row16
ME_HD int read_off( int c, int row16, int strip_elems )
{
    return ( c >> 3 ) * strip_elems + ( c & 7 ) + row16;
}
// the two taps of the quarter-pel sample run at (mvx, mvy) quarter-pels, as offsets into the strips of the FOUR half-pel planes (plane
// p's strips start at p * 2 * plane_elems).  o = read_off() of the full-pel part of the vector: ( cx0 + ( mvx >> 2 ), row16 + 16 *
// ( mvy >> 2 ) ).  The plane pair of each of the 16 phases comes from two 32-bit lookup constants (two bits per phase:
// pa = (fx ? 1 : 0) + (fy == 2 ? 2 : 0), pb = (fx == 2 ? 1 : 0) + (fy ? 2 : 0), common/mc.c:198-212's x264_hpel_ref0 / ref1); the
// partner column / row is +1 / +16 samples inside the same strip.
ME_HD void qpel_taps( int plane_elems, int o, int mvx, int mvy, int &oa, int &ob )
{
    const int fx = mvx & 3, fy = mvy & 3;
    const int sh = 2 * ( fx | ( fy << 2 ) );
    const unsigned pa = ( 0x54FE5454u >> sh ) & 3u, pb = ( 0xBABABA10u >> sh ) & 3u;
    oa = (int)( pa * (unsigned)plane_elems * 2u ) + o + ( fy == 3 ? 16 : 0 );
    ob = (int)( pb * (unsigned)plane_elems * 2u ) + o + ( fx == 3 );
}
}
