// shard_host.cpp -- ONE lookahead window over the GPUs of a node, callable from C: the orchestration of SURVEY 8(e) / BASELINE configs[3]
// inside libx264hip.so, so that the reference's C host (encoder/lookahead.c:90-128, the thread that calls x264_slicetype_decide) can use
// every GPU of the box the way it uses one -- the shape of the accelerator precedent it already has, encoder/slicetype-cl.h:29-42.
//
// Rank 0 drives an ordinary host lookahead (x264hip_lookahead_put_frame / _get_frame on x264hip_shard_lookahead()); wherever that would
// submit speculative work for a chunk of frames it broadcasts the chunk instead, and every rank derives the same plan from it: frame
// b's unweighted searches and cost cells belong to rank b % world.  Per chunk, on each rank, everything enqueued on the context's own
// stream (collectives included, no host thread waits):
//   0. the chunk's new pictures are broadcast (W x H samples each); every rank makes its own lowres planes;
//   1. the searches of its frames;
//   2. the list-0 fields B cells on OTHER ranks read from their list-1 reference (encoder/slicetype.c:629-642) go peer to peer in
//      exact counts;
//   3. the cost cells of its frames (B cells both ways where the reference's field exists);
//   4. their summaries (X264HIP_CELL_SUMMARY_INTS ints per cell) are gathered on rank 0, which registers the other ranks' fields as
//      searched elsewhere and takes the summaries in as speculative cells;
//   5. a status word is max-reduced over the ranks: a rank whose device calls failed keeps issuing the collectives of the plan (the
//      counts are known to everybody) with whatever its buffers hold, reports the failure there, and every rank stops at the next
//      command -- a failing rank fails the call on every rank instead of leaving the others inside a collective.
// Per-block maps stay with their owners; before an MB-tree call rank 0 names the maps that call reads in a FETCH command.
// Every exchange step goes through buffers allocated ONCE, at open (pictures, fields, summaries, maps: a fixed number of bytes each);
// what does not fit travels in pieces whose sizes every rank derives from the plan.  Nothing is allocated between a command and
// its collectives, so no rank can drop out of a collective because an allocation failed (round-5 review).
// The exchange goes through an x264hip_shard_transport: RCCL over xGMI (x264hip_shard_transport_rccl: librccl is dlopen'ed, the
// library does not link it) or anything with the same four operations (the tests run two ranks on one GPU over a host-staged one).
// Results are those of the single stream by construction: a search is a pure function of its two frames.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <set>
#include <tuple>
#include <vector>

#include "x264hip.h"

namespace
{
enum { CMD_STOP = 0, CMD_CHUNK = 1, CMD_FETCH = 2, CMD_RESET = 3 };
const int CMD_WORDS = 8192; // int64 words per command: a chunk names <= 2 * (frames of a chunk + the window) + (bframes+2)^2 of them

typedef std::tuple<int, int, int> Key3; // (frame number, list, dist - 1) or (frame number, d0, d1)
} // namespace

struct x264hip_shard
{
    x264hip_shard_transport T;
    int rank = 0, world = 1, device = 0, loopback = 0;
    x264hip_lookahead *la = nullptr;
    x264hip_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    int bframes = 0, ns = 0, n_mb = 0, mb_h = 0, width = 0, height = 0, pix = 1;
    int failed = 0;            // first error of this rank (X264HIP_E*); once set the rank only keeps the collectives going
    int stop_seen = 0;
    int *status_dev = nullptr; // [2]: this command's status word, reduced over the ranks
    int *status_host = nullptr;
    hipEvent_t status_ev = nullptr;
    int status_posted = 0;
    static const int CMD_RING = 8;
    int64_t *cmd_dev = nullptr, *cmd_host = nullptr; // cmd_host: CMD_RING pinned blocks (rank 0 writes them round robin; the others use the first)
    hipEvent_t cmd_ev[8] = { nullptr };
    int cmd_ev_used[8] = { 0 }, cmd_next = 0;
    std::set<Key3> done, cells_done;
    std::set<int> sums_done, ingested;
    std::set<std::tuple<int, int, int>> l0_sent; // (receiving rank, frame number, dist - 1)
    std::map<int, int> slot_of, number_in;        // frame number -> slot, slot -> frame number
    std::map<int, const void *> picture;          // rank 0: frame number -> device pointer of its luma (x264hip_shard_put_frames)
    int picture_stride = 0;
    // the exchange buffers (device memory, allocated at open, reused by every command in stream order)
    char *pic_buf = nullptr;                    // pictures: pic_frames frames per piece
    char *l0_sbuf = nullptr, *l0_rbuf = nullptr; // list-0 fields: l0_fields fields per peer and piece, world peers
    char *sum_sbuf = nullptr, *sum_rbuf = nullptr; // cell summaries: sum_cells cells per rank and piece (rbuf: rank 0, world blocks)
    char *map_sbuf = nullptr, *map_rbuf = nullptr; // cell maps: map_cells per rank and piece
    size_t pic_frames = 0, l0_fields = 0, sum_cells = 0, map_cells = 0;
    uint64_t stats[X264HIP_SHARD_STATS] = { 0 };
    std::vector<std::pair<void *, void *>> loop_checks; // loopback: (sent, received) pairs compared at close
    std::vector<size_t> loop_bytes;

    int fail( int rc ) { if( rc && !failed ) failed = rc; return rc; }
    // loop-back runs keep a copy of what was sent and of what came back (the exchange buffers are reused), compared at close
    void loop_keep( const void *sent, const void *got, size_t bytes )
    {
        void *a = nullptr, *b = nullptr;
        if( !bytes || hipMalloc( &a, bytes ) != hipSuccess || hipMalloc( &b, bytes ) != hipSuccess ||
            hipMemcpyAsync( a, sent, bytes, hipMemcpyDeviceToDevice, stream ) != hipSuccess ||
            hipMemcpyAsync( b, got, bytes, hipMemcpyDeviceToDevice, stream ) != hipSuccess )
        {
            (void)hipGetLastError(); (void)hipFree( a ); (void)hipFree( b ); // (a check that cannot be kept is skipped: one rank, nobody waits)
            return;
        }
        loop_checks.push_back( std::make_pair( a, b ) ); loop_bytes.push_back( bytes );
    }
    // entries of frames no later command can name: everything older than the oldest frame of a chunk by more than the reach of a reference
    void prune( int oldest )
    {
        const int limit = oldest - bframes - 1;
        done.erase( done.begin(), done.lower_bound( Key3( limit, -1, -1 ) ) );
        cells_done.erase( cells_done.begin(), cells_done.lower_bound( Key3( limit, -1, -1 ) ) );
        sums_done.erase( sums_done.begin(), sums_done.lower_bound( limit ) );
        ingested.erase( ingested.begin(), ingested.lower_bound( limit ) );
        picture.erase( picture.begin(), picture.lower_bound( limit ) );
        for( auto it = l0_sent.begin(); it != l0_sent.end(); )
            if( std::get<1>( *it ) < limit ) it = l0_sent.erase( it ); else ++it;
        for( auto it = slot_of.begin(); it != slot_of.end() && it->first < limit; )
        {
            auto in = number_in.find( it->second );
            if( in != number_in.end() && in->second == it->first ) number_in.erase( in );
            it = slot_of.erase( it );
        }
    }
    void forget_sequence() // a new sequence starts at frame 0 (x264hip_shard_reset)
    {
        done.clear(); cells_done.clear(); sums_done.clear(); ingested.clear(); l0_sent.clear(); slot_of.clear(); number_in.clear(); picture.clear();
    }
};

namespace
{
int owner( const x264hip_shard *s, int number ) { return ( ( number % s->world ) + s->world ) % s->world; }

// ---- commands: a fixed-size block of int64 words from rank 0 to everybody (word 0 = words used) --------------------------------------
int send_cmd( x264hip_shard *s, const std::vector<int64_t> &cmd )
{
    if( s->world == 1 && !s->loopback ) return X264HIP_OK;
    if( (int)cmd.size() + 1 > CMD_WORDS ) return s->fail( X264HIP_EINVAL );
    // (the pinned block is read when the copy runs, not when it is enqueued: a ring, each entry reused only after its copy is done)
    const int k = s->cmd_next++ % x264hip_shard::CMD_RING;
    if( s->cmd_ev_used[k] && hipEventSynchronize( s->cmd_ev[k] ) != hipSuccess ) return s->fail( X264HIP_EDEVICE );
    int64_t *h = s->cmd_host + (size_t)k * CMD_WORDS;
    h[0] = (int64_t)cmd.size();
    memcpy( h + 1, cmd.data(), cmd.size() * sizeof( int64_t ) );
    const size_t bytes = ( cmd.size() + 1 ) * sizeof( int64_t );
    if( hipMemcpyAsync( s->cmd_dev, h, bytes, hipMemcpyHostToDevice, s->stream ) != hipSuccess || hipEventRecord( s->cmd_ev[k], s->stream ) != hipSuccess )
        return s->fail( X264HIP_EDEVICE );
    s->cmd_ev_used[k] = 1;
    // (the block travels whole: the receivers do not know its length before they have it)
    if( s->T.broadcast( s->T.user, s->cmd_dev, CMD_WORDS * sizeof( int64_t ), 0, s->stream ) ) return s->fail( X264HIP_EDEVICE );
    return X264HIP_OK;
}
int recv_cmd( x264hip_shard *s, std::vector<int64_t> &cmd )
{
    if( s->T.broadcast( s->T.user, s->cmd_dev, CMD_WORDS * sizeof( int64_t ), 0, s->stream ) ) return s->fail( X264HIP_EDEVICE );
    if( hipMemcpyAsync( s->cmd_host, s->cmd_dev, CMD_WORDS * sizeof( int64_t ), hipMemcpyDeviceToHost, s->stream ) != hipSuccess ||
        hipStreamSynchronize( s->stream ) != hipSuccess )
        return s->fail( X264HIP_EDEVICE );
    const int64_t n = s->cmd_host[0];
    if( n < 1 || n + 1 > CMD_WORDS ) return s->fail( X264HIP_ESTATE );
    cmd.assign( s->cmd_host + 1, s->cmd_host + 1 + n );
    return X264HIP_OK;
}

// the status word of a command: every rank contributes its own, everybody learns the worst.  Enqueued behind the command's work and
// looked at when it has arrived (a failed rank keeps reporting in every later command, so the latest word says it all): no host thread
// waits for a chunk it does not need yet.  wait: block until the last word posted is in (close, x264hip_shard_status, serve).
int post_status( x264hip_shard *s )
{
    if( s->world == 1 && !s->loopback ) return X264HIP_OK;
    // error codes are negative: the maximum of the negated codes is the worst
    if( hipMemsetD32Async( (hipDeviceptr_t)s->status_dev, s->failed ? -s->failed : 0, 1, s->stream ) != hipSuccess ) return s->fail( X264HIP_EDEVICE );
    if( s->T.allreduce_max_i32( s->T.user, s->status_dev, 1, s->stream ) ) return s->fail( X264HIP_EDEVICE );
    if( hipMemcpyAsync( s->status_host + 1, s->status_dev, sizeof( int ), hipMemcpyDeviceToHost, s->stream ) != hipSuccess ||
        hipEventRecord( s->status_ev, s->stream ) != hipSuccess )
        return s->fail( X264HIP_EDEVICE );
    s->status_posted = 1;
    return X264HIP_OK;
}
int check_status( x264hip_shard *s, bool wait )
{
    if( !s->status_posted ) return s->failed;
    if( wait ? hipEventSynchronize( s->status_ev ) == hipSuccess : hipEventQuery( s->status_ev ) == hipSuccess )
    {
        if( s->status_host[1] > 0 && !s->failed ) s->failed = X264HIP_EPEER;
    }
    else if( wait )
        s->fail( X264HIP_EDEVICE );
    return s->failed;
}

struct Field { int slot_b, slot_ref, list, dm1, number; };
struct Cell { int slot_b, slot_p0, slot_p1, d0, d1, flags, number; };

// ---- the plan of a chunk: identical on every rank (x264_amd/shard.py WindowShard.plan) ------------------------------------------------
void plan( x264hip_shard *s, const std::vector<int> &slots, const std::vector<int> &numbers, unsigned m0, unsigned m1, const std::vector<int> &cell_class,
           std::vector<std::vector<Field>> &fields, std::vector<std::vector<Cell>> &cells, std::vector<std::set<std::pair<int, int>>> &l0_wanted )
{
    const int bf = s->bframes, ns = s->ns, n = (int)numbers.size();
    fields.assign( s->world, {} ); cells.assign( s->world, {} ); l0_wanted.assign( s->world, {} );
    const unsigned masks[2] = { m0, m1 };
    for( int i = 0; i < n; i++ )
        for( int j = 0; j < n; j++ )
        {
            const int d = numbers[j] - numbers[i];
            if( !d || abs( d ) > bf + 1 || ( d > 0 && !bf ) ) continue;
            const int lst = d > 0, dm1 = abs( d ) - 1;
            if( !( ( masks[lst] >> dm1 ) & 1 ) || s->done.count( Key3( numbers[i], lst, dm1 ) ) ) continue;
            s->done.insert( Key3( numbers[i], lst, dm1 ) );
            fields[owner( s, numbers[i] )].push_back( Field{ slots[i], slots[j], lst, dm1, numbers[i] } );
        }
    std::map<int, int> here; // a chunk names every resident frame the decisions can reach: only those are safe to refer to
    for( int i = 0; i < n; i++ ) here[numbers[i]] = slots[i];
    for( int i = 0; i < n; i++ )
    {
        const int ni = numbers[i];
        for( int d0 = 1; d0 <= bf + 1; d0++ )
        {
            const int p0 = ni - d0;
            if( !here.count( p0 ) || !s->done.count( Key3( ni, 0, d0 - 1 ) ) ) continue;
            for( int d1 = 0; d0 + d1 <= bf + 1; d1++ )
            {
                const int c = cell_class[d0 * ns + d1];
                if( !c || s->cells_done.count( Key3( ni, d0, d1 ) ) ) continue;
                int p1 = ni, with_l0 = 0;
                if( d1 )
                {
                    p1 = ni + d1; with_l0 = c == 2;
                    if( !here.count( p1 ) || !s->done.count( Key3( ni, 1, d1 - 1 ) ) ) continue;
                    const bool ref_field = s->done.count( Key3( p1, 0, d0 + d1 - 1 ) ) != 0;
                    if( c == 3 ) with_l0 = ref_field ? ( X264HIP_CELL_WITH_L0 | X264HIP_CELL_BOTH ) : 0;
                    if( with_l0 && !ref_field ) continue;
                    if( with_l0 && ( owner( s, p1 ) != owner( s, ni ) || s->loopback ) )
                        l0_wanted[owner( s, ni )].insert( std::make_pair( p1, d0 + d1 - 1 ) );
                }
                s->cells_done.insert( Key3( ni, d0, d1 ) );
                cells[owner( s, ni )].push_back( Cell{ slots[i], here[p0], here[p1], d0, d1, with_l0, ni } );
            }
        }
    }
}

x264hip_cell_ref ref_of( const Cell &c, int flags ) { x264hip_cell_ref r = { c.slot_b, c.slot_p0, c.slot_p1, c.d0, c.d1, flags }; return r; }
// summaries: a cell evaluated both ways travels as two entries, its own (flag WITH_L0) and its spare half (flag SPARE)
std::vector<x264hip_cell_ref> halves( const std::vector<Cell> &cs )
{
    std::vector<x264hip_cell_ref> out;
    for( const Cell &c : cs )
        if( c.flags & X264HIP_CELL_BOTH ) { out.push_back( ref_of( c, X264HIP_CELL_WITH_L0 ) ); out.push_back( ref_of( c, X264HIP_CELL_SPARE ) ); }
        else out.push_back( ref_of( c, c.flags ) );
    return out;
}

int run_chunk( x264hip_shard *s, const std::vector<int> &slots, const std::vector<int> &numbers, unsigned m0, unsigned m1, const std::vector<int> &cell_class )
{
    const bool exchanging = s->world > 1 || s->loopback;
    const size_t frame_bytes = (size_t)s->width * s->height * s->pix;
    {
        // fault injection for the tests of the error path: X264HIP_SHARD_TEST_FAIL="<rank>:<chunk>" makes that rank fail at the start of
        // that chunk (as if a device call had failed there)
        static const char *inj = getenv( "X264HIP_SHARD_TEST_FAIL" );
        int r = -1, c = -1;
        if( inj && sscanf( inj, "%d:%d", &r, &c ) == 2 && r == s->rank && (uint64_t)c == s->stats[X264HIP_SHARD_CHUNKS] )
            s->fail( X264HIP_EDEVICE );
    }
    if( !numbers.empty() )
        s->prune( *std::min_element( numbers.begin(), numbers.end() ) );
    // ---- 0. the pictures of the chunk's new frames: from rank 0's own buffers through the picture buffer of every other rank,
    //         pic_frames at a time (the buffer is reused in stream order: a piece's ingest kernels run before the next broadcast lands)
    if( exchanging )
    {
        std::vector<int> fresh;
        for( int nmb : numbers ) if( !s->ingested.count( nmb ) ) fresh.push_back( nmb );
        std::sort( fresh.begin(), fresh.end() );
        for( int nmb : fresh ) s->ingested.insert( nmb );
        for( size_t first = 0; first < fresh.size(); first += s->pic_frames )
        {
            const size_t cnt = std::min( s->pic_frames, fresh.size() - first );
            if( s->rank == 0 )
                for( size_t k = 0; k < cnt; k++ )
                {
                    auto it = s->picture.find( fresh[first + k] );
                    if( it == s->picture.end() ) { s->fail( X264HIP_ESTATE ); continue; }
                    // (rows packed: the block is what travels)
                    if( hipMemcpy2DAsync( s->pic_buf + k * frame_bytes, (size_t)s->width * s->pix, it->second, (size_t)s->picture_stride * s->pix, (size_t)s->width * s->pix,
                                          s->height, hipMemcpyDeviceToDevice, s->stream ) != hipSuccess )
                        s->fail( X264HIP_EDEVICE );
                }
            if( s->T.broadcast( s->T.user, s->pic_buf, frame_bytes * cnt, 0, s->stream ) ) s->fail( X264HIP_EDEVICE );
            s->stats[X264HIP_SHARD_BYTES_INPUT] += s->world > 1 ? frame_bytes * cnt : 0;
            if( s->rank )
                for( size_t k = 0; k < cnt; k++ )
                {
                    // the slot rank 0 announced for this frame
                    int slot = -1;
                    for( size_t i = 0; i < numbers.size(); i++ ) if( numbers[i] == fresh[first + k] ) slot = slots[i];
                    if( slot < 0 ) continue;
                    if( !s->failed )
                        s->fail( x264hip_frame_put( s->ctx, slot, s->pic_buf + k * frame_bytes, s->width, 1, nullptr, nullptr, 0, nullptr ) );
                }
            if( s->loopback && !s->failed ) // the picture that came back is the one rank 0 holds
            {
                s->loop_checks.push_back( std::make_pair( (void *)nullptr, (void *)nullptr ) ); s->loop_bytes.push_back( 0 );
            }
        }
    }
    for( size_t i = 0; i < slots.size(); i++ )
    {
        auto it = s->number_in.find( slots[i] );
        if( it == s->number_in.end() || it->second != numbers[i] )
        {
            if( it != s->number_in.end() ) s->slot_of.erase( it->second );
            s->number_in[slots[i]] = numbers[i];
            s->slot_of[numbers[i]] = slots[i]; // (a frame number never comes back in another slot within a stream)
        }
    }
    std::vector<std::vector<Field>> fields;
    std::vector<std::vector<Cell>> cells;
    std::vector<std::set<std::pair<int, int>>> l0_wanted;
    plan( s, slots, numbers, m0, m1, cell_class, fields, cells, l0_wanted );
    // ---- 1. the searches of this rank's frames
    {
        const std::vector<Field> &mine = fields[s->rank];
        if( !mine.empty() && !s->failed )
        {
            std::vector<int> a, b, c, d;
            for( const Field &f : mine ) { a.push_back( f.slot_b ); b.push_back( f.slot_ref ); c.push_back( f.list ); d.push_back( f.dm1 ); }
            s->fail( x264hip_search_fields( s->ctx, (int)mine.size(), a.data(), b.data(), c.data(), d.data() ) );
        }
        s->stats[X264HIP_SHARD_CHUNKS]++;
        s->stats[X264HIP_SHARD_FIELDS_SEARCHED] += mine.size();
    }
    // ---- 2. list-0 fields for B cells on other ranks: every rank sends each peer exactly the fields that peer asked for, at most
    //         l0_fields per pair and piece (piece i of a pair = its fields [i * l0_fields, (i+1) * l0_fields): every rank knows every list)
    if( exchanging )
    {
        const size_t field_bytes = (size_t)s->n_mb * 2 * sizeof( int );
        // give[o][r]: fields rank o owns that rank r wants, in a fixed order
        std::vector<std::vector<std::vector<std::pair<int, int>>>> give( s->world, std::vector<std::vector<std::pair<int, int>>>( s->world ) );
        size_t longest = 0;
        for( int r = 0; r < s->world; r++ )
            for( const auto &k : l0_wanted[r] )
            {
                if( s->l0_sent.count( std::make_tuple( r, k.first, k.second ) ) ) continue;
                s->l0_sent.insert( std::make_tuple( r, k.first, k.second ) );
                std::vector<std::pair<int, int>> &g = give[owner( s, k.first )][r];
                g.push_back( k ); // (std::set iteration is sorted: the same order on every rank)
                longest = std::max( longest, g.size() );
            }
        for( size_t first = 0; first < longest; first += s->l0_fields )
        {
            auto piece = [&]( const std::vector<std::pair<int, int>> &g ) { return g.size() > first ? std::min( s->l0_fields, g.size() - first ) : (size_t)0; };
            std::vector<size_t> sb( s->world ), rb( s->world );
            size_t n_send = 0, n_recv = 0, row = 0;
            for( int r = 0; r < s->world; r++ )
            {
                sb[r] = piece( give[s->rank][r] ) * field_bytes; rb[r] = piece( give[r][s->rank] ) * field_bytes;
                n_send += piece( give[s->rank][r] ); n_recv += piece( give[r][s->rank] );
            }
            for( int r = 0; r < s->world; r++ )
                for( size_t k = 0; k < piece( give[s->rank][r] ); k++, row++ )
                {
                    const std::pair<int, int> &f = give[s->rank][r][first + k];
                    auto it = s->slot_of.find( f.first );
                    if( !s->failed && it != s->slot_of.end() )
                        s->fail( x264hip_export_field( s->ctx, it->second, 0, f.second, s->l0_sbuf + row * field_bytes ) );
                }
            if( s->T.send_recv( s->T.user, s->l0_sbuf, sb.data(), s->l0_rbuf, rb.data(), s->stream ) ) s->fail( X264HIP_EDEVICE );
            row = 0;
            for( int o = 0; o < s->world; o++ )
                for( size_t k = 0; k < piece( give[o][s->rank] ); k++, row++ )
                {
                    const std::pair<int, int> &f = give[o][s->rank][first + k];
                    auto it = s->slot_of.find( f.first );
                    if( !s->failed && it != s->slot_of.end() )
                        s->fail( x264hip_import_field( s->ctx, it->second, 0, f.second, s->l0_rbuf + row * field_bytes ) );
                    s->stats[X264HIP_SHARD_L0_FIELDS]++;
                }
            s->stats[X264HIP_SHARD_BYTES_L0] += n_recv * field_bytes;
            if( s->loopback && n_send && !s->failed )
                s->loop_keep( s->l0_sbuf, s->l0_rbuf, n_send * field_bytes );
        }
    }
    // ---- 3. the cells of the frames this rank owns; rank 0 also queues the intra sums of every frame (it has all of them resident)
    {
        std::vector<x264hip_cell_ref> list;
        if( s->rank == 0 )
            for( size_t i = 0; i < slots.size(); i++ )
                if( !s->sums_done.count( numbers[i] ) )
                {
                    s->sums_done.insert( numbers[i] );
                    x264hip_cell_ref r = { slots[i], slots[i], slots[i], 0, 0, 0 };
                    list.push_back( r );
                }
        for( const Cell &c : cells[s->rank] ) list.push_back( ref_of( c, c.flags ) );
        if( !list.empty() && !s->failed )
            s->fail( x264hip_spec_cells( s->ctx, (int)list.size(), list.data() ) );
        s->stats[X264HIP_SHARD_CELLS_EVALUATED] += cells[s->rank].size();
    }
    if( !exchanging ) return s->failed;
    // ---- 4. summaries to rank 0, sum_cells entries per rank and piece
    {
        std::vector<std::vector<x264hip_cell_ref>> sent( s->world );
        size_t wide = 0;
        for( int r = 0; r < s->world; r++ )
        {
            sent[r] = halves( cells[r] );
            if( r || s->loopback ) wide = std::max( wide, sent[r].size() );
        }
        std::vector<int> rs, rn, rl, rd;
        if( s->rank == 0 )
            for( int r = 1; r < s->world; r++ )
                for( const Field &f : fields[r] ) { rs.push_back( f.slot_b ); rn.push_back( f.number ); rl.push_back( f.list ); rd.push_back( f.dm1 ); }
        if( s->rank == 0 && !rs.empty() && !s->failed )
            s->fail( x264hip_fields_remote( s->ctx, (int)rs.size(), rs.data(), rn.data(), rl.data(), rd.data() ) );
        const size_t per = X264HIP_CELL_SUMMARY_INTS( s->mb_h ) * sizeof( int );
        for( size_t first = 0; first < wide; first += s->sum_cells )
        {
            const size_t cnt = std::min( s->sum_cells, wide - first ), bytes = cnt * per;
            auto piece = [&]( int r ) { return sent[r].size() > first ? std::min( cnt, sent[r].size() - first ) : (size_t)0; };
            if( hipMemsetAsync( s->sum_sbuf, 0, bytes, s->stream ) != hipSuccess ) s->fail( X264HIP_EDEVICE );
            if( piece( s->rank ) && ( s->rank || s->loopback ) && !s->failed )
                s->fail( x264hip_export_cells( s->ctx, (int)piece( s->rank ), sent[s->rank].data() + first, s->sum_sbuf ) );
            if( s->T.gather( s->T.user, s->sum_sbuf, s->sum_rbuf, bytes, 0, s->stream ) ) s->fail( X264HIP_EDEVICE );
            if( s->rank == 0 )
            {
                for( int r = s->loopback ? 0 : 1; r < s->world; r++ )
                    if( piece( r ) && !s->failed ) // (loopback: every entry is skipped -- the cells are here)
                        s->fail( x264hip_import_cells( s->ctx, (int)piece( r ), sent[r].data() + first, s->sum_rbuf + (size_t)r * bytes ) );
                s->stats[X264HIP_SHARD_BYTES_SUMMARIES] += (size_t)( s->world - 1 ) * bytes;
            }
            if( s->loopback && piece( 0 ) && !s->failed )
                s->loop_keep( s->sum_sbuf, s->sum_rbuf, piece( 0 ) * per );
        }
        if( s->rank == 0 )
            for( int r = s->loopback ? 0 : 1; r < s->world; r++ )
                if( !sent[r].empty() ) s->stats[X264HIP_SHARD_CELLS_IMPORTED] += cells[r].size();
    }
    // ---- 5. how it went, for everybody
    post_status( s );
    return s->failed;
}

int fetch_maps( x264hip_shard *s, const std::vector<x264hip_cell_ref> &cells, const std::vector<int> &numbers )
{
    std::vector<std::vector<x264hip_cell_ref>> by_owner( s->world );
    for( size_t k = 0; k < cells.size(); k++ ) by_owner[owner( s, numbers[k] )].push_back( cells[k] );
    size_t wide = 0;
    for( int r = 1; r < s->world; r++ ) wide = std::max( wide, by_owner[r].size() );
    if( !wide ) return s->failed;
    const size_t map_bytes = (size_t)3 * s->n_mb * sizeof( int );
    for( size_t first = 0; first < wide; first += s->map_cells )
    {
        const size_t cnt = std::min( s->map_cells, wide - first ), bytes = cnt * map_bytes;
        auto piece = [&]( int r ) { return by_owner[r].size() > first ? std::min( cnt, by_owner[r].size() - first ) : (size_t)0; };
        if( s->rank )
            for( size_t k = 0; k < piece( s->rank ); k++ )
                if( !s->failed )
                    s->fail( x264hip_export_cell_map( s->ctx, &by_owner[s->rank][first + k], s->map_sbuf + k * map_bytes ) );
        if( s->T.gather( s->T.user, s->map_sbuf, s->map_rbuf, bytes, 0, s->stream ) ) s->fail( X264HIP_EDEVICE );
        if( s->rank == 0 )
        {
            for( int r = 1; r < s->world; r++ )
                for( size_t k = 0; k < piece( r ); k++ )
                {
                    x264hip_cell_ref c = by_owner[r][first + k];
                    c.with_ref1_l0 = 0;
                    if( !s->failed )
                        s->fail( x264hip_import_cell_map( s->ctx, &c, s->map_rbuf + (size_t)r * bytes + k * map_bytes ) );
                    s->stats[X264HIP_SHARD_MAPS_FETCHED]++;
                }
            s->stats[X264HIP_SHARD_BYTES_MAPS] += (size_t)( s->world - 1 ) * bytes;
        }
    }
    if( s->rank == 0 ) s->stats[X264HIP_SHARD_FETCH_COMMANDS]++;
    post_status( s );
    return s->failed;
}

// ---- rank 0: the host lookahead's two hooks -------------------------------------------------------------------------------------------
int on_prefetch( void *user, const int *slots, const int *numbers, int n )
{
    x264hip_shard *s = (x264hip_shard *)user;
    if( check_status( s, false ) ) return s->failed;
    unsigned m0 = 0, m1 = 0;
    std::vector<unsigned char> cc( (size_t)s->ns * s->ns );
    if( s->fail( x264hip_field_classes( s->ctx, &m0, &m1 ) ) || s->fail( x264hip_cell_classes( s->ctx, cc.data() ) ) ) return s->failed;
    std::vector<int64_t> cmd = { CMD_CHUNK, n, (int64_t)m0, (int64_t)m1 };
    for( int i = 0; i < n; i++ ) cmd.push_back( slots[i] );
    for( int i = 0; i < n; i++ ) cmd.push_back( numbers[i] );
    for( unsigned char v : cc ) cmd.push_back( v );
    if( send_cmd( s, cmd ) ) return s->failed;
    std::vector<int> cls( cc.begin(), cc.end() );
    return run_chunk( s, std::vector<int>( slots, slots + n ), std::vector<int>( numbers, numbers + n ), m0, m1, cls );
}

int before_mbtree( void *user, const x264hip_mbtree_op *ops, int n )
{
    x264hip_shard *s = (x264hip_shard *)user;
    if( s->world == 1 ) return X264HIP_OK;
    if( check_status( s, false ) ) return s->failed;
    std::vector<x264hip_cell_ref> cells;
    std::set<std::tuple<int, int, int, int, int>> seen;
    for( int i = 0; i < n; i++ )
        if( ops[i].type == X264HIP_MBT_PROPAGATE && seen.insert( std::make_tuple( ops[i].slot_b, ops[i].slot_p0, ops[i].slot_p1, ops[i].dist_p0, ops[i].dist_p1 ) ).second )
        {
            x264hip_cell_ref r = { ops[i].slot_b, ops[i].slot_p0, ops[i].slot_p1, ops[i].dist_p0, ops[i].dist_p1, 0 };
            cells.push_back( r );
        }
    if( cells.empty() ) return X264HIP_OK;
    std::vector<unsigned char> missing( cells.size() );
    if( s->fail( x264hip_cells_missing( s->ctx, (int)cells.size(), cells.data(), missing.data() ) ) ) return s->failed;
    std::vector<x264hip_cell_ref> miss;
    std::vector<int> numbers;
    for( size_t k = 0; k < cells.size(); k++ )
        if( missing[k] )
        {
            x264hip_cell_ref c = cells[k];
            c.with_ref1_l0 = missing[k] == 2 ? X264HIP_CELL_SPARE : 0; // the half of the cell that is wanted
            if( missing[k] == 2 ) s->stats[X264HIP_SHARD_MAPS_FETCHED_SPARE]++;
            miss.push_back( c ); numbers.push_back( s->number_in[c.slot_b] );
        }
    if( miss.empty() ) return X264HIP_OK;
    std::vector<int64_t> cmd = { CMD_FETCH, (int64_t)miss.size() };
    for( size_t k = 0; k < miss.size(); k++ )
    {
        const x264hip_cell_ref &c = miss[k];
        for( int v : { c.slot_b, c.slot_p0, c.slot_p1, c.dist_p0, c.dist_p1, c.with_ref1_l0, numbers[k] } ) cmd.push_back( v );
    }
    if( send_cmd( s, cmd ) ) return s->failed;
    return fetch_maps( s, miss, numbers );
}
} // namespace

// ---- public entry points -----------------------------------------------------------------------------------------------------------------
namespace
{
// what x264hip_shard_open created, and nothing else: the transport stays the caller's until an open has succeeded
void free_own( x264hip_shard *s )
{
    (void)hipSetDevice( s->device );
    if( s->stream ) (void)hipStreamSynchronize( s->stream );
    for( auto &c : s->loop_checks ) { if( c.first ) (void)hipFree( c.first ); if( c.second ) (void)hipFree( c.second ); }
    if( s->la ) x264hip_lookahead_close( s->la );
    if( s->status_ev ) (void)hipEventDestroy( s->status_ev );
    for( int k = 0; k < x264hip_shard::CMD_RING; k++ ) if( s->cmd_ev[k] ) (void)hipEventDestroy( s->cmd_ev[k] );
    (void)hipFree( s->status_dev ); (void)hipHostFree( s->status_host ); (void)hipFree( s->cmd_dev ); (void)hipHostFree( s->cmd_host );
    for( char *b : { s->pic_buf, s->l0_sbuf, s->l0_rbuf, s->sum_sbuf, s->sum_rbuf, s->map_sbuf, s->map_rbuf } ) (void)hipFree( b );
    delete s;
}
// bytes per exchange buffer: X264HIP_SHARD_PIECE_BYTES (the tests make it small, so that every step travels in several pieces)
size_t piece_budget( size_t dflt )
{
    const char *e = getenv( "X264HIP_SHARD_PIECE_BYTES" );
    const long long v = e ? atoll( e ) : 0;
    return v > 0 ? (size_t)v : dflt;
}
} // namespace

extern "C" int x264hip_shard_open( x264hip_shard **out, int device, const x264hip_la_params *params, const x264hip_shard_transport *transport )
{
    if( !out || !params || !transport || transport->world < 1 || transport->rank < 0 || transport->rank >= transport->world ) return X264HIP_EINVAL;
    if( !transport->broadcast || !transport->send_recv || !transport->gather || !transport->allreduce_max_i32 ) return X264HIP_EINVAL;
    *out = nullptr;
    x264hip_shard *s = new x264hip_shard;
    s->T = *transport; s->rank = transport->rank; s->world = transport->world; s->device = device;
    s->loopback = s->world == 1 && transport->loopback;
    const bool hooked = s->world > 1 || s->loopback;
    int rc = s->rank == 0 && hooked ? x264hip_lookahead_open_hooked( &s->la, device, params, on_prefetch, s ) : x264hip_lookahead_open( &s->la, device, params );
    if( rc ) { delete s; return rc; }
    s->ctx = x264hip_lookahead_ctx( s->la );
    void *st = nullptr;
    rc = x264hip_stream_handle( s->ctx, &st );
    s->stream = (hipStream_t)st;
    int mb_w = 0, stride = 0;
    if( !rc ) rc = x264hip_geometry( s->ctx, &mb_w, &s->mb_h, &stride );
    s->n_mb = mb_w * s->mb_h;
    s->bframes = params->dev.bframes; s->ns = s->bframes + 2;
    s->width = params->dev.width; s->height = params->dev.height; s->pix = params->dev.bit_depth > 8 ? 2 : 1;
    if( !rc && s->rank == 0 && s->world > 1 ) rc = x264hip_lookahead_set_mbtree_hook( s->la, before_mbtree, s );
    if( !rc && ( hipSetDevice( device ) != hipSuccess || hipMalloc( &s->status_dev, 2 * sizeof( int ) ) != hipSuccess ||
                 hipHostMalloc( &s->status_host, 2 * sizeof( int ) ) != hipSuccess || hipEventCreateWithFlags( &s->status_ev, hipEventDisableTiming ) != hipSuccess || hipMalloc( &s->cmd_dev, CMD_WORDS * sizeof( int64_t ) ) != hipSuccess ||
                 hipHostMalloc( &s->cmd_host, (size_t)x264hip_shard::CMD_RING * CMD_WORDS * sizeof( int64_t ) ) != hipSuccess ) )
        rc = X264HIP_ENOMEM;
    for( int k = 0; k < x264hip_shard::CMD_RING && !rc; k++ )
        if( hipEventCreateWithFlags( &s->cmd_ev[k], hipEventDisableTiming ) != hipSuccess ) rc = X264HIP_ENOMEM;
    if( !rc && hooked )
    {
        // the exchange buffers: whole frames / fields / summaries / maps per piece, at least one
        const size_t frame_bytes = (size_t)s->width * s->height * s->pix, field_bytes = (size_t)s->n_mb * 2 * sizeof( int ),
                     sum_bytes = X264HIP_CELL_SUMMARY_INTS( s->mb_h ) * sizeof( int ), map_bytes = (size_t)3 * s->n_mb * sizeof( int );
        s->pic_frames = std::max<size_t>( 1, piece_budget( (size_t)128 << 20 ) / frame_bytes );
        s->l0_fields = std::max<size_t>( 1, piece_budget( (size_t)32 << 20 ) / ( field_bytes * s->world ) );
        s->sum_cells = std::max<size_t>( 1, piece_budget( (size_t)4 << 20 ) / sum_bytes );
        s->map_cells = std::max<size_t>( 1, piece_budget( (size_t)32 << 20 ) / map_bytes );
        const size_t gathered = s->rank == 0 ? s->world : 0;
        if( hipMalloc( &s->pic_buf, s->pic_frames * frame_bytes ) != hipSuccess ||
            hipMalloc( &s->l0_sbuf, s->l0_fields * s->world * field_bytes ) != hipSuccess || hipMalloc( &s->l0_rbuf, s->l0_fields * s->world * field_bytes ) != hipSuccess ||
            hipMalloc( &s->sum_sbuf, s->sum_cells * sum_bytes ) != hipSuccess || ( gathered && hipMalloc( &s->sum_rbuf, s->sum_cells * sum_bytes * gathered ) != hipSuccess ) ||
            hipMalloc( &s->map_sbuf, s->map_cells * map_bytes ) != hipSuccess || ( gathered && hipMalloc( &s->map_rbuf, s->map_cells * map_bytes * gathered ) != hipSuccess ) )
        {
            (void)hipGetLastError();
            rc = X264HIP_ENOMEM;
        }
    }
    if( rc ) { free_own( s ); return rc; } // (no STOP, no transport destroy: nothing was agreed with anybody yet)
    s->status_host[0] = s->status_host[1] = 0;
    *out = s;
    return X264HIP_OK;
}

extern "C" x264hip_lookahead *x264hip_shard_lookahead( x264hip_shard *s ) { return s ? s->la : nullptr; }
extern "C" x264hip_ctx *x264hip_shard_ctx( x264hip_shard *s ) { return s ? s->ctx : nullptr; }

// rank 0: the pictures of frames first_number .. first_number + n - 1 (device pointers, `stride` samples per row), which must stay valid
// until the chunk that names them has been submitted (x264hip_lookahead_get_frame consumed them); then put into the lookahead
extern "C" int x264hip_shard_put_frames( x264hip_shard *s, int first_number, int n, const void *const *luma_dev, int stride )
{
    if( !s || s->rank != 0 || n < 0 || ( n && !luma_dev ) ) return X264HIP_EINVAL;
    if( s->failed ) return s->failed;
    for( int i = 0; i < n; i++ ) s->picture[first_number + i] = luma_dev[i];
    s->picture_stride = stride;
    return x264hip_lookahead_put_frames( s->la, n, luma_dev, stride );
}

// ranks 1 .. world-1: execute rank 0's commands until it closes its shard (or somebody fails); returns X264HIP_OK or the first error
extern "C" int x264hip_shard_serve( x264hip_shard *s )
{
    if( !s || s->rank == 0 ) return X264HIP_EINVAL;
    if( hipSetDevice( s->device ) != hipSuccess ) return X264HIP_EDEVICE;
    while( true )
    {
        std::vector<int64_t> cmd;
        if( recv_cmd( s, cmd ) ) return s->failed; // (the transport itself failed: nothing more can be agreed on)
        check_status( s, true ); // the previous command's verdict (recv_cmd waited for the stream: it is in)
        if( cmd[0] == CMD_STOP ) { s->stop_seen = 1; return s->failed; }
        if( cmd[0] == CMD_RESET )
        {
            s->forget_sequence();
            if( !s->failed ) s->fail( x264hip_lookahead_reset( s->la ) );
            post_status( s );
            continue;
        }
        if( cmd[0] == CMD_FETCH )
        {
            // (the block is the same on every rank: a record count that does not match its length is rank 0's mistake, and rank 0 --
            // which built it -- sees the same thing; the status word says so, nobody enters a gather the others size differently)
            const int64_t n = cmd.size() >= 2 ? cmd[1] : -1;
            if( n < 0 || (int64_t)cmd.size() != 2 + 7 * n ) { s->fail( X264HIP_ESTATE ); post_status( s ); continue; }
            std::vector<x264hip_cell_ref> cells;
            std::vector<int> numbers;
            for( int k = 0; k < (int)n; k++ )
            {
                const int64_t *r = &cmd[2 + 7 * k];
                x264hip_cell_ref c = { (int)r[0], (int)r[1], (int)r[2], (int)r[3], (int)r[4], (int)r[5] };
                cells.push_back( c ); numbers.push_back( (int)r[6] );
            }
            fetch_maps( s, cells, numbers );
            continue;
        }
        if( cmd[0] != CMD_CHUNK || cmd.size() < 4 ) { s->fail( X264HIP_ESTATE ); post_status( s ); continue; }
        const int n = (int)cmd[1], ns2 = s->ns * s->ns;
        if( n < 0 || (int)cmd.size() < 4 + 2 * n + ns2 ) { s->fail( X264HIP_ESTATE ); post_status( s ); continue; }
        std::vector<int> slots( n ), numbers( n ), cls( ns2 );
        for( int i = 0; i < n; i++ ) { slots[i] = (int)cmd[4 + i]; numbers[i] = (int)cmd[4 + n + i]; }
        for( int i = 0; i < ns2; i++ ) cls[i] = (int)cmd[4 + 2 * n + i];
        run_chunk( s, slots, numbers, (unsigned)cmd[2], (unsigned)cmd[3], cls );
    }
}

// everything enqueued so far has run; X264HIP_OK, this rank's first error, or X264HIP_EPEER when another rank reported one
extern "C" int x264hip_shard_status( x264hip_shard *s )
{
    if( !s ) return X264HIP_EINVAL;
    if( hipSetDevice( s->device ) != hipSuccess || hipStreamSynchronize( s->stream ) != hipSuccess ) return s->fail( X264HIP_EDEVICE );
    return check_status( s, true );
}

extern "C" int x264hip_shard_stats( x264hip_shard *s, uint64_t *out, int n )
{
    if( !s || !out ) return X264HIP_EINVAL;
    for( int i = 0; i < n && i < X264HIP_SHARD_STATS; i++ ) out[i] = s->stats[i];
    return X264HIP_OK;
}

// loopback (a one-rank transport with loopback set): what came back from every exchange equals what the export kernels wrote
extern "C" int x264hip_shard_loopback_verify( x264hip_shard *s, int *n_checked )
{
    if( !s ) return X264HIP_EINVAL;
    if( hipStreamSynchronize( s->stream ) != hipSuccess ) return X264HIP_EDEVICE;
    int checked = 0, bad = 0;
    for( size_t k = 0; k < s->loop_checks.size(); k++ )
    {
        if( !s->loop_checks[k].first ) continue;
        std::vector<char> a( s->loop_bytes[k] ), b( s->loop_bytes[k] );
        if( hipMemcpy( a.data(), s->loop_checks[k].first, a.size(), hipMemcpyDeviceToHost ) != hipSuccess ||
            hipMemcpy( b.data(), s->loop_checks[k].second, b.size(), hipMemcpyDeviceToHost ) != hipSuccess )
            return X264HIP_EDEVICE;
        bad += memcmp( a.data(), b.data(), a.size() ) != 0;
        checked++;
        (void)hipFree( s->loop_checks[k].first ); (void)hipFree( s->loop_checks[k].second );
    }
    s->loop_checks.clear(); s->loop_bytes.clear();
    if( n_checked ) *n_checked = checked;
    return bad ? X264HIP_ESTATE : X264HIP_OK;
}

// A new sequence on every rank (frame numbers start at 0 again): rank 0 calls this INSTEAD of x264hip_lookahead_reset on the shard's
// lookahead -- the ranks key what they hold by frame number, so resetting rank 0's lookahead alone would make them take the new
// sequence's frames for ones they have already ingested and searched.
extern "C" int x264hip_shard_reset( x264hip_shard *s )
{
    if( !s || s->rank != 0 ) return X264HIP_EINVAL;
    if( check_status( s, false ) ) return s->failed;
    if( send_cmd( s, std::vector<int64_t>( 1, CMD_RESET ) ) ) return s->failed;
    s->forget_sequence();
    s->fail( x264hip_lookahead_reset( s->la ) );
    post_status( s );
    return s->failed;
}

extern "C" void x264hip_shard_close( x264hip_shard *s )
{
    if( !s ) return;
    (void)hipSetDevice( s->device );
    if( s->rank == 0 && s->world > 1 && s->cmd_dev && s->stream )
        send_cmd( s, std::vector<int64_t>( 1, CMD_STOP ) ); // also after a failure: the other ranks are waiting for a command
    x264hip_shard_transport T = s->T;
    free_own( s );
    if( T.destroy ) T.destroy( T.user );
}

// ---- the RCCL transport: librccl.so is opened at run time (the library, like the reference's OpenCL path, common/opencl.c:53-61, builds and
// loads without it) -----------------------------------------------------------------------------------------------------------------------------
namespace
{
typedef struct { char internal[128]; } NcclId;
typedef void *NcclComm;
struct Rccl
{
    void *lib = nullptr;
    int ( *GetUniqueId )( NcclId * ) = nullptr;
    int ( *CommInitRank )( NcclComm *, int, NcclId, int ) = nullptr;
    int ( *CommDestroy )( NcclComm ) = nullptr;
    int ( *Broadcast )( const void *, void *, size_t, int, int, NcclComm, hipStream_t ) = nullptr;
    int ( *AllReduce )( const void *, void *, size_t, int, int, NcclComm, hipStream_t ) = nullptr;
    int ( *Send )( const void *, size_t, int, int, NcclComm, hipStream_t ) = nullptr;
    int ( *Recv )( void *, size_t, int, int, NcclComm, hipStream_t ) = nullptr;
    int ( *GroupStart )() = nullptr;
    int ( *GroupEnd )() = nullptr;
};
Rccl *rccl()
{
    static Rccl R;
    static bool tried = false;
    if( tried ) return R.lib ? &R : nullptr;
    tried = true;
    const char *names[] = { getenv( "X264HIP_RCCL_LIB" ), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
    for( const char *nm : names )
        if( nm && ( R.lib = dlopen( nm, RTLD_NOW | RTLD_GLOBAL ) ) ) break;
    if( !R.lib ) return nullptr;
#define SYM( f, name ) *(void **)&R.f = dlsym( R.lib, name )
    SYM( GetUniqueId, "ncclGetUniqueId" ); SYM( CommInitRank, "ncclCommInitRank" ); SYM( CommDestroy, "ncclCommDestroy" ); SYM( Broadcast, "ncclBroadcast" );
    SYM( AllReduce, "ncclAllReduce" ); SYM( Send, "ncclSend" ); SYM( Recv, "ncclRecv" ); SYM( GroupStart, "ncclGroupStart" ); SYM( GroupEnd, "ncclGroupEnd" );
#undef SYM
    if( !R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.Broadcast || !R.AllReduce || !R.Send || !R.Recv || !R.GroupStart || !R.GroupEnd )
    {
        dlclose( R.lib ); R.lib = nullptr;
        return nullptr;
    }
    return &R;
}
struct RcclT { NcclComm comm; int rank, world; };
enum { NCCL_INT8 = 0, NCCL_INT32 = 2, NCCL_MAX = 2 }; // ncclDataType_t / ncclRedOp_t (nccl.h: ncclInt8 0, ncclInt32 2; ncclSum 0, ncclProd 1, ncclMax 2)

int rccl_broadcast( void *u, void *buf, size_t bytes, int root, void *stream )
{
    RcclT *t = (RcclT *)u;
    return rccl()->Broadcast( buf, buf, bytes, NCCL_INT8, root, t->comm, (hipStream_t)stream ) ? -1 : 0;
}
int rccl_send_recv( void *u, const void *sbuf, const size_t *sb, void *rbuf, const size_t *rb, void *stream )
{
    RcclT *t = (RcclT *)u;
    Rccl *R = rccl();
    int bad = R->GroupStart();
    size_t so = 0, ro = 0;
    for( int r = 0; r < t->world; r++ )
    {
        if( sb[r] ) bad |= R->Send( (const char *)sbuf + so, sb[r], NCCL_INT8, r, t->comm, (hipStream_t)stream );
        if( rb[r] ) bad |= R->Recv( (char *)rbuf + ro, rb[r], NCCL_INT8, r, t->comm, (hipStream_t)stream );
        so += sb[r]; ro += rb[r];
    }
    bad |= R->GroupEnd();
    return bad ? -1 : 0;
}
int rccl_gather( void *u, const void *sbuf, void *rbuf, size_t bytes, int root, void *stream )
{
    RcclT *t = (RcclT *)u;
    Rccl *R = rccl();
    int bad = R->GroupStart();
    bad |= R->Send( sbuf, bytes, NCCL_INT8, root, t->comm, (hipStream_t)stream );
    if( t->rank == root )
        for( int r = 0; r < t->world; r++ )
            bad |= R->Recv( (char *)rbuf + (size_t)r * bytes, bytes, NCCL_INT8, r, t->comm, (hipStream_t)stream );
    bad |= R->GroupEnd();
    return bad ? -1 : 0;
}
int rccl_allreduce_max( void *u, int *buf, int n, void *stream )
{
    RcclT *t = (RcclT *)u;
    return rccl()->AllReduce( buf, buf, (size_t)n, NCCL_INT32, NCCL_MAX, t->comm, (hipStream_t)stream ) ? -1 : 0;
}
void rccl_destroy( void *u )
{
    RcclT *t = (RcclT *)u;
    if( t && t->comm ) rccl()->CommDestroy( t->comm );
    delete t;
}
} // namespace

extern "C" int x264hip_rccl_unique_id( void *id128 )
{
    Rccl *R = rccl();
    if( !R || !id128 ) return R ? X264HIP_EINVAL : X264HIP_ENODEV;
    NcclId id;
    if( R->GetUniqueId( &id ) ) return X264HIP_EDEVICE;
    memcpy( id128, &id, sizeof( id ) );
    return X264HIP_OK;
}

// one communicator per shard: every rank calls this with the id rank 0 obtained from x264hip_rccl_unique_id (and handed round by the
// caller's own means -- a file, a pipe, MPI, the environment) after hipSetDevice( device ).  world == 1 gives a loop-back communicator.
extern "C" int x264hip_shard_transport_rccl( x264hip_shard_transport *t, const void *nccl_unique_id, int rank, int world, int device )
{
    Rccl *R = rccl();
    if( !t || !nccl_unique_id || world < 1 || rank < 0 || rank >= world ) return X264HIP_EINVAL;
    if( !R ) return X264HIP_ENODEV;
    if( hipSetDevice( device ) != hipSuccess ) return X264HIP_EDEVICE;
    NcclId id;
    memcpy( &id, nccl_unique_id, sizeof( id ) );
    RcclT *u = new RcclT;
    u->comm = nullptr; u->rank = rank; u->world = world;
    if( R->CommInitRank( &u->comm, world, id, rank ) ) { delete u; return X264HIP_EDEVICE; }
    memset( t, 0, sizeof( *t ) );
    t->user = u; t->rank = rank; t->world = world;
    t->broadcast = rccl_broadcast; t->send_recv = rccl_send_recv; t->gather = rccl_gather; t->allreduce_max_i32 = rccl_allreduce_max; t->destroy = rccl_destroy;
    return X264HIP_OK;
}
