// kernels.hip — gfx950 (MI355X / CDNA4) kernels of libteloscan.
//
// ts_scan_tiles: the telomeric-motif scan of Teloscope::scanSegment / analyzeWindow
// (reference src/teloscope.cpp:387-658) for uniform-length pattern sets, as one persistent
// kernel.  Integer/byte work bounded by HBM (1 B/base in, 32 B/window + 4 B/match out);
// no MFMA on purpose.
//
// Execution model: ONE WAVEFRONT PER TILE.  A tile is a run of consecutive windows of one
// segment (up to ~16 k bases incl. the w-s halo); a wave scans its tile start to finish out of its own
// slice of LDS, and the waves of a workgroup share nothing but the read-only match table.
// After the table is loaded there is no workgroup barrier and no wait on another wave anywhere: a wave takes
// its next tile from a ticket counter (one relaxed global atomic per tile, issued a tile ahead of its use; tiles
// are dealt round-robin instead where per-wave record counts must be reproducible), and every wave appends its packed
// match records to its own region of the output (a per-tile directory {offset, count} makes
// the stream addressable in position order).  Latency is hidden by the 16 waves per CU — or 20, as two workgroups of
// ten, where the pattern set is dense enough to need them (the WAVES_EU = 6 build below; plan_geometry in capi.cpp decides) —
// and by wave priorities: the phases that are chains of dependent LDS / memory operations run above the dense per-chunk work.
//
// Per tile:
//   phase 1  2016 positions per iteration, 32 per lane: two coalesced 16 B/lane loads (the next
//            chunk's in flight; a tile's first chunk is requested while the previous tile's window
//            phase runs), SWAR ASCII -> 2-bit codes (v_and / v_perm / v_bitop3, v_dot4 packs), ONE LDS read per
//            TWO positions from the pair table ((k+1)-mer -> match bits of both positions; one byte
//            per (k+1)-mer for k <= 6, so the (k+1)-mer is the address); the codes go to the tile's code plane.
//            The chunk's matches (a few per cent of the positions) are compacted onto a queue and
//            resolved 64 at a time on full wavefronts: forward/canonical flags from the flag
//            table, the packed 32-bit record (staged in LDS, flushed in coalesced rows), and one
//            packed ds_add_u64 per window that contains the match.
//   phase 2  nucleotide fields of the windows, counted from the code plane (three popcounts per 32 positions,
//            per step block when w is a multiple of s); 8 x u32 per window leave coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ts_internal.h"

// Profiling only: -DTS_ABL=<mask> builds a kernel with one stage removed (results are then wrong)
// so that stage costs can be measured under real overlap; see profiles/ablate.sh.
//   1 record flush to global  2 nucleotide window sums  4 window match accumulation
//   8 window record stores  16 the whole per-match pass  32 table probes  64 code plane stores
//   128 the consumption of the match queue (compaction still runs)
#ifndef TS_ABL
#define TS_ABL 0
#endif
// Experiments and diagnostics (profiles/abx.sh): -DTS_EXP=<mask>.  8: every tile leaves the 100 MHz timestamp of its end
// (20 bits) and the wave that scanned it (12 bits) in the spare word of its tile_stats row (profiles/tile_timeline.py
// reads the waves' timelines from them).
// 1: no wave priorities (every phase at priority 0, as before they were introduced).
#ifndef TS_EXP
#define TS_EXP 0
#endif
// 1: the emitting build requests its next tile at the start of phase 2 instead of ahead of the tile's last passes (-0.4 % on configs[1]: profiles/r05/emit_ab_plain_emit_r04_r05_late_ffirst.txt; 0 for A/B)
// whole rows of 64 records take a copy of the row code without the mask of live lanes (0: one copy for every row)
#ifndef TS_EMIT_FULL_ROWS
#define TS_EMIT_FULL_ROWS 1
#endif
#ifndef TS_EMIT_LATE_REQUEST
#define TS_EMIT_LATE_REQUEST 1
#endif
// (measurement: results are then wrong) 1: the rows are not looked at, 2: no visible records, 4: no chain summary
#ifndef TS_EMIT_ABL
#define TS_EMIT_ABL 0
#endif
#ifndef TS_EMIT_FINISH_LAST
#define TS_EMIT_FINISH_LAST 1
#endif
#ifndef TS_PLAIN_FINISH_LAST
#define TS_PLAIN_FINISH_LAST 0
#endif

// Wave priorities (s_setprio; the SIMD's arbiter picks the ready wave of the highest priority, the oldest among equals).
// The dense per-chunk work (decode, probes: long runs of independent vector instructions) stays at 0; the phases that are
// chains of dependent LDS / memory operations with a few instructions between the waits run above it, so that those few
// instructions issue as soon as their operand arrives instead of queueing behind another wave's dense run, and the wave
// is back in dense work sooner: configs[1] -2.8 .. -3.8 %, the k = 7 and default-flag configurations -1.5 %
// (profiles/r02/kernel_experiments_ab.txt, "wave priorities").
#ifndef TS_PRIO_PASS
#define TS_PRIO_PASS 3          // the per-match pass over 64 queued matches
#endif
#ifndef TS_PRIO_APPEND
#define TS_PRIO_APPEND 2        // prefix sum of a chunk pair's match counts + the queue writes
#endif
#ifndef TS_PRIO_WINDOWS
#define TS_PRIO_WINDOWS 2       // phase 2: record flush, nucleotide counts, window records, tile directory
#endif
#define set_prio(p) do { if (!(TS_EXP & 1)) __builtin_amdgcn_s_setprio(p); } while (0)
constexpr int kPrioPass = TS_PRIO_PASS, kPrioAppend = TS_PRIO_APPEND, kPrioWindows = TS_PRIO_WINDOWS;

namespace {

typedef unsigned long long u64;
// LDS pointers are declared in their own address space so that indexing stays 32-bit arithmetic
#define LDS __attribute__((address_space(3)))
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef LDS unsigned char lds_u8;
typedef LDS uint16_t lds_u16;
typedef LDS uint32_t lds_u32;

// an LDS byte address (the dynamic-LDS base is 0) as a pointer
__device__ __forceinline__ lds_u16 *lds_at16(uint32_t addr) { return (lds_u16 *)(uintptr_t)addr; }
__device__ __forceinline__ lds_u8 *lds_at8(uint32_t addr) { return (lds_u8 *)(uintptr_t)addr; }

// Global stores the compiler does not see (TS_ASM_STORES, default on).  On gfx9 loads and stores share one counter (vmcnt) and may
// retire out of order with respect to each other, so once a store is pending the compiler can only wait for a LOAD with vmcnt(0) —
// and it does so early: in front of every loop that holds a store and no load it empties the counter ("flush in the preheader").
// In this kernel that meant an s_waitcnt vmcnt(0) right behind the request of the next tile's first chunk (the last drain of the
// match queue is such a loop: its overflow path stores), i.e. the prefetch was waited for on the spot, and two more in phase 2.
// The stores never feed a load of this kernel (the one place that reads records back waits for vmcnt(0) itself), so they are issued
// by inline asm: the compiler counts only its loads, whose waits stay counted, and a pending store can only make such a wait longer,
// never too short (loads retire in order among themselves).  Measurements: profiles/r05/asm_stores.txt.
// -DTS_READ_INDEX_BUILD=1: the emitting build also knows kp.emit == 2 (canonical-record indices for ts_read_predicate_canon, the read
// filter experiment of profiles/r05/reads_canonical_index_experiment.txt); without it ts_k_read_index_built() says no and
// TS_READ_EMIT=1 is ignored.
#ifndef TS_READ_INDEX_BUILD
#define TS_READ_INDEX_BUILD 0
#endif
#ifndef TS_ASM_STORES
#define TS_ASM_STORES 1
#endif
__device__ __forceinline__ void gstore(uint32_t *p, uint32_t v) {
#if TS_ASM_STORES
    asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void gstore(uint16_t *p, uint16_t v) {
#if TS_ASM_STORES
    asm volatile("global_store_short %0, %1, off" :: "v"(p), "v"((uint32_t)v));
#else
    *p = v;
#endif
}
// (the low 16 bits of v: the store takes them itself)
__device__ __forceinline__ void gstore_lo16(uint16_t *p, uint32_t v) {
#if TS_ASM_STORES
    asm volatile("global_store_short %0, %1, off" :: "v"(p), "v"(v));
#else
    *p = (uint16_t)v;
#endif
}
__device__ __forceinline__ void gstore(unsigned char *p, unsigned char v) {
#if TS_ASM_STORES
    asm volatile("global_store_byte %0, %1, off" :: "v"(p), "v"((uint32_t)v));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void gstore(u64 *p, u64 v) {
#if TS_ASM_STORES
    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void gstore(uint4 *p, uint4 v) {
#if TS_ASM_STORES
    const u32x4 d = {v.x, v.y, v.z, v.w};
    // (s_nop: a store of more than 8 bytes reads its data a cycle late, and the hazard recogniser does not look into asm)
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 0" :: "v"(p), "v"(d));
#else
    *p = v;
#endif
}
// (possibly unaligned: the bit-packed window records)
__device__ __forceinline__ void gstore_unaligned(unsigned char *p, u64 v) {
#if TS_ASM_STORES
    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v));
#else
    __builtin_memcpy(p, &v, 8);
#endif
}
__device__ __forceinline__ void gstore_unaligned(unsigned char *p, uint32_t v) {
#if TS_ASM_STORES
    asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v));
#else
    __builtin_memcpy(p, &v, 4);
#endif
}

// Wave-wide inclusive prefix sum in 6 DPP adds (row_shr 1/2/4/8 inside each row of 16,
// then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3); no LDS traffic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);
    v = dpp_add<0x143, 0xc>(v);
    return v;
}

// (a DPP move folded into the subtraction that follows it came back wrong on gfx950 — see blockcall.hip — so the places that take the
// lane below's value keep it a v_mov_b32_dpp behind an empty asm)
__device__ __forceinline__ unsigned long long low_bits(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }

// the lanes for which `p` holds, as a mask: on a bool this is one scalar AND of the compare's result with exec (__ballot takes
// an int: the bool is first materialised per lane and compared again, two vector instructions per ballot)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }


// sixteen 2-bit codes, held doubled (2 * code = ASCII & 6) one per byte of t[0..3] -> one dword, base i at
// bits 2i..2i+1; four independent v_dot4_u32_u8 (weights 1,4,16,64: twice the packed byte, 9 bits) and four
// shift/or, no dependent dot chain
__device__ __forceinline__ uint32_t pack16(const uint32_t t[4]) {
    const uint32_t b0 = __builtin_amdgcn_udot4(t[0], 0x40100401u, 0u, false);
    const uint32_t b1 = __builtin_amdgcn_udot4(t[1], 0x40100401u, 0u, false);
    const uint32_t b2 = __builtin_amdgcn_udot4(t[2], 0x40100401u, 0u, false);
    const uint32_t b3 = __builtin_amdgcn_udot4(t[3], 0x40100401u, 0u, false);
    return ((b0 | (b1 << 8)) >> 1) | ((b2 | (b3 << 8)) << 15);
}


__host__ __device__ inline uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }

// LDS: [match table][wave 0 queue][wave 1 queue]...[wave 0 slice][wave 1 slice]...; a slice = codes | rec | wacc | stage
//   codes   2-bit codes of the tile, two dwords per 32 positions (+ one look-ahead dword pair); the per-match pass
//           reads k-mers from it and the window phase counts the nucleotides of every step block in it
//   queue   (its own region) ring of TS_LIST u16 plane coordinates of match positions, appended to by every
//           chunk in position order and consumed 64 at a time, so that the per-match work runs
//           on full wavefronts
//   rec     nucleotide counts {A, C, G, T}, 4 x u32 per step block of the tile (windows + halo) when w is a
//           multiple of s (a window is the sum of w / s rows), else per window
//   wacc    the match fields of the tile's window records while they accumulate: one u64 per step block (w a multiple of s:
//           a window's fields are then the sum of its blocks' minus the matches that run over the end of its last block,
//           which have a row of their own — ONE add per match instead of one per window that contains it) or per window
//           = four 16-bit counters {canonical, non-canonical, forward, reverse}, bumped by ONE
//           ds_add_u64 per (match, window); kept in acc_copies lane-interleaved copies so that the
//           matches of a pass, which mostly fall into the same few windows, do not serialise on one
//           address
//   stage   packed match records waiting to leave in whole coalesced rows at the tile's end (kept out of the
//           chunk loads' counted vmcnt waits), 16 bits each when tile positions allow (P.stage_u16), + one spare slot per lane
//           for predicated-off writes
//   park    eight dwords of per-wave state of the emitting build that only the end of a tile touches (finish_records): kept here,
//           not in scalar registers, so that the chunk loop of the emitting build is the plain build's
struct SliceLayout { uint32_t codes, rec, wacc, stage, park, bytes; };

__host__ __device__ inline SliceLayout slice_layout(const TsScanParams &P) {
    SliceLayout s;
    uint32_t o = 0;
    s.codes = o; o += align16((P.nch * 63u + 1u) * 8u);
    s.rec = o; o += (P.windows_on && P.nuc_on) ? align16((P.max_windows + P.halo_blocks) * 16u) : 0u;
    // (per step block when w is a multiple of s: rows for the tile's windows + halo in every copy, then one row per block for
    //  the matches that run over their block's end; else per window)
    s.wacc = o; o += P.windows_on ? align16((P.max_windows + (P.acc_blocks ? P.halo_blocks : 0u)) * 8u * (P.acc_copies + (P.acc_blocks ? 1u : 0u))) : 0u;
    s.stage = o; o += (P.stage_cap + 64u) * (P.stage_u16 ? 2u : 4u);
    s.park = o; o += 32u;                                     // (whether or not this batch emits: one geometry)
    s.bytes = o;
    return s;
}

// The match queues of all waves sit together behind the table, each aligned to its own size, so that a
// queue slot's LDS address is (byte offset & mask) | base: one v_and_or instead of mask, shift and add.
__host__ __device__ inline uint32_t queue_region(const TsScanParams &P) {
    const uint32_t a = TS_LIST * 2u;
    return (P.table_rows * 4u + P.fc_bytes + a - 1u) & ~(a - 1u);
}

__host__ __device__ inline uint32_t lds_total(const TsScanParams &P) {
    return queue_region(P) + P.waves_per_wg * (TS_LIST * 2u + slice_layout(P).bytes);
}

// ---------------------------------------------------------------------------------------
// Two register budgets.  WAVES_EU = 4: one workgroup of up to 16 waves per CU (4 per SIMD; the allocator takes 81 VGPRs).
// WAVES_EU = 6: at most 80 VGPRs, so that TWO workgroups of 10 waves share a CU — a workgroup's waves go round the SIMDs
// 3,3,2,2 and two of them put six on SIMD 0, which 81 registers (five waves per SIMD) do not allow: the second workgroup then
// waits for the first (measured: 1.14 ms instead of 0.74).
// EMIT: the build that also leaves the visible records and the per-tile chain summaries (P.emit).  A build of its own, so that a
// scan nobody calls blocks from afterwards (a resident scan whose results stay in HBM) runs the code it always ran: with the
// emit state merely branched around, the register allocator spilled 60 more scalars in the chunk loop (+7 % on configs[1]).
// FAST: the build for what nearly every scan is — 16-bit stage entries (tiles of at most eight chunks) and, in a window scan, match
// fields accumulated per step block (w a multiple of s) — with both decided at compile time: the other forms' code and the
// registers it keeps alive through the chunk loop are then not there (byte tables only: the configurations that are timed).
template <bool FC_BYTES, bool PAIR_BYTES, int WAVES_EU, bool EMIT, bool FAST>
__global__ __launch_bounds__(TS_MAX_WG_THREADS, WAVES_EU)
void ts_scan_tiles(const TsScanParams P) {
    const bool st16 = FAST ? true : P.stage_u16 != 0u;
    const bool accb = FAST ? true : P.acc_blocks != 0u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));

    // the replicated bit table is loaded once and stays for the lifetime of the workgroup
    const uint32_t table_bytes = P.table_rows * 4u + P.fc_bytes;                  // pair table + flag table
    {
        const uint4 *src = (const uint4 *)P.table;
        uint4 *dst = (uint4 *)lds_raw;
        const uint32_t n16 = table_bytes >> 4;
        for (uint32_t i = tid; i < n16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();                              // the only workgroup barrier in the kernel

    const SliceLayout SL = slice_layout(P);
    lds_u8 *lds = (lds_u8 *)lds_raw;
    uint32_t qbase = queue_region(P) + wave * (TS_LIST * 2u);                    // LDS byte address of this wave's queue
    uint32_t qmask = TS_LIST * 2u - 2u;
    asm volatile("" : "+v"(qbase));               // base in a VGPR, mask in an SGPR: (offset & mask) | base is one v_and_or_b32
    asm volatile("" : "+s"(qmask));
    lds_u8 *slice = lds + queue_region(P) + P.waves_per_wg * (TS_LIST * 2u) + wave * SL.bytes;
    lds_u32 *codes = (lds_u32 *)(slice + SL.codes);
    lds_u32 *rec = (lds_u32 *)(slice + SL.rec);
    lds_u32 *stage = (lds_u32 *)(slice + SL.stage);
    LDS u64 *wacc = (LDS u64 *)(slice + SL.wacc);
    lds_u32 *park = (lds_u32 *)(slice + SL.park);          // EMIT: [4] the wave's visible cursor, [5] "this tile flushed from inside a pass" (finish_records)
    const uint32_t acc_rows = P.max_windows + (accb ? P.halo_blocks : 0u);   // rows of one copy of wacc
    const uint32_t acc_off = (lane & (P.acc_copies - 1u)) * acc_rows;          // this lane's copy of wacc

    const uint32_t k = P.k;
    const uint32_t rowbits = 2u * (k + 1u) - 4u;      // pair table: 4^(k+1) entries, 16 per dword
    const uint32_t kmask = (1u << (2u * k)) - 1u;
    const uint32_t pmask = (1u << (2u * (k + 1u))) - 1u;              // a (k+1)-mer = an index into the byte pair table
    const lds_u32 *fc_table = (const lds_u32 *)(lds + P.table_rows * 4u);
    const lds_u8 *fc_bytes = (const lds_u8 *)fc_table;
    // LDS byte address of the table (it sits at the dynamic-LDS base, which is 0: the kernel has no static LDS)
    const uint32_t tab_base = (uint32_t)(uintptr_t)lds;
    if (PAIR_BYTES && (tab_base & pmask) != 0u) __builtin_trap();     // the byte-table probes OR the index into the base

    const uint32_t total_waves = gridDim.x * P.waves_per_wg;
    const uint32_t gw = blockIdx.x * P.waves_per_wg + wave;
    // Parameters that are only needed when a tile is finished (output pointers, capacities) are read
    // from the kernel-argument segment (scalar loads, constant address space) where they are used,
    // through a pointer the compiler cannot see through: held in SGPRs for the whole kernel they push the
    // hot loops' scalars into spills (v_readlane / v_writelane are VALU instructions).
    typedef const TsScanParams __attribute__((address_space(4))) *KernArgs;
    auto tail_params = [&]() -> KernArgs {
        KernArgs q = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(q));
        return q;
    };
    uint32_t cursor = 0;                          // records this wave has produced so far
    if (EMIT) { park[4] = 0u; park[5] = 0u; }     // [4] visible records this wave has produced so far, [5] tile mark (every lane writes the same words)
    const uint32_t nwper = P.halo_blocks + 1u;             // windows a position can belong to: ceil(w / s)

    // The first chunk of a tile is fetched while the previous tile's window phase runs (its loads would
    // otherwise be waited for with nothing else to do): descriptor and 32 B/lane of the next tile are
    // requested at the end of phase 1 and picked up here.
    //
    // Which tile a wave scans next.  Dealt round-robin (tile = gw, gw + waves, ...) the waves of a launch do not finish
    // together: those that share a SIMD are not served evenly and a tile inside a telomere costs more than one outside —
    // on configs[1] the fastest wave was done after 66 % of the kernel's time and the median one after 80 %
    // (profiles/r02/tile_timeline_dealt.txt), so the last fifth of the kernel ran on half-empty SIMDs.  With
    // P.dynamic_tiles a wave TAKES its next tile: the workgroups form P.ticket_groups groups (workgroup % groups), group g
    // owns the tiles t = g (mod groups) and hands them out in order through a ticket counter (one relaxed atomic add per
    // tile, by lane 0).  Several counters because same-address atomics are served one at a time at the memory side
    // (~7 ns each: one counter for all 222 k tiles of configs[1] made the kernel 2.7x slower); interleaved ownership so
    // that every group sees the same mix of contigs.  The ticket is taken a whole tile ahead — the atomic for the tile
    // after next is issued where the next tile's first chunk is requested and its result is needed one phase 1 later —
    // so no wait is spent on it.  Without P.dynamic_tiles the per-wave record counts are reproducible, which the sizing
    // of small batches and the rescan after an overflow rely on.  The counters of the NEXT launch are zeroed here (two
    // sets used alternately: launches of one batch are ordered on its stream).
    // (tile descriptors are written by the host only: read through the constant address space they come by scalar loads — as
    // plain global loads they came into vector registers, seven v_readfirstlane per tile, behind a vmcnt wait)
    typedef const TsTile __attribute__((address_space(4))) *ConstTiles;
    const ConstTiles const_tiles = (ConstTiles)(uintptr_t)P.tiles;
    uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;
    uint32_t ticket = 0;                          // lane 0: tickets taken in the group before this one
    const uint32_t ngroups = P.dynamic_tiles ? P.ticket_groups : 1u;
    const uint32_t group = blockIdx.x % ngroups;
    const uint32_t group_waves = ((gridDim.x - group + ngroups - 1u) / ngroups) * P.waves_per_wg;
    // The counter's address goes through a VGPR pair the compiler cannot see through: for an atomic on a uniform
    // address in divergent code it would otherwise build its wave-reduction form, which waits for the returned
    // value on the spot (s_waitcnt vmcnt(0) behind the next tile's loads).
    auto take_ticket = [&](KernArgs Q) -> uint32_t {
        const u64 a = (u64)(uintptr_t)(Q->tile_tickets + (Q->ticket_slot * Q->ticket_groups + group) * TS_TICKET_STRIDE);
        uint32_t lo = (uint32_t)a, hi = (uint32_t)(a >> 32);
        asm volatile("" : "+v"(lo), "+v"(hi));
        typedef __attribute__((address_space(1))) uint32_t *GlobalU32;       // global, not flat: a flat atomic also counts on lgkmcnt
        return __hip_atomic_fetch_add((GlobalU32)(uintptr_t)(((u64)hi << 32) | lo), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (P.dynamic_tiles && gw == 0 && lane < ngroups) P.tile_tickets[((1u - P.ticket_slot) * ngroups + lane) * TS_TICKET_STRIDE] = 0u;
    // dealt: tile = gw + i * waves; taken: tile = group + groups * (index in the group's order), the wave's first index
    // being its own number in the group
    const uint32_t first_tile = P.dynamic_tiles ? group + ngroups * ((blockIdx.x / ngroups) * P.waves_per_wg + wave) : gw;
    if (first_tile < P.ntiles) {
        if (P.dynamic_tiles && lane == 0) ticket = take_ticket(tail_params());
        const unsigned char *ls = P.in + (const_tiles[first_tile].in_off & ~15ull) + lane * 32u;
        n0 = *(const uint4 *)ls; n1 = *(const uint4 *)(ls + 16);
    }
    for (uint32_t tile = first_tile, tile_next = 0; tile < P.ntiles; tile = tile_next) {
        // wave-uniform, by scalar loads (the request for the tile's first chunk read its in_off a phase ago: not kept)
        // Only what phase 1 needs is kept, each field a scalar of its own (as one eight-register tuple the descriptor was
        // spilled and reloaded whole — eight moves — wherever one field of it was wanted); the index of the tile's first window
        // record is read again where phase 2 needs it.
        struct { u64 in_off; uint32_t nrel, nwin, own_len; } T;
        {
            const ConstTiles q = const_tiles + tile;
            uint32_t lo = (uint32_t)q->in_off, hi = (uint32_t)(q->in_off >> 32), a = q->nrel, b = q->nwin, c = q->own_len;
            asm volatile("" : "+s"(lo), "+s"(hi), "+s"(a), "+s"(b), "+s"(c));
            T.in_off = ((u64)hi << 32) | lo; T.nrel = a; T.nwin = b; T.own_len = c;
        }
        const uint32_t sh = (uint32_t)(T.in_off & 15ull);
        const unsigned char *src = P.in + (T.in_off - sh);
        const uint32_t nblk = T.nwin + P.halo_blocks;             // step blocks the tile's windows reach into
        const uint32_t span = nblk * P.s;
        const uint32_t xend = sh + T.nrel;                         // plane coord of the segment end
        uint32_t need = sh + (T.nrel < span + 16u ? T.nrel : span + 16u);
        uint32_t nch = (need + 16u + TS_CHUNK - 1u) / TS_CHUNK;
        if (nch > P.nch) nch = P.nch;
        nch = (uint32_t)__builtin_amdgcn_readfirstlane((int)nch);
        bool has_invalid = false;
        const uint32_t own_end = sh + T.own_len;                   // plane coord: positions [sh, own_end) are this tile's
        uint32_t done = 0, ccan = 0, cfwd = 0;                     // records of this tile so far, canonical / forward among them (uniform)
        uint32_t flushed = 0;                                      // how many of them have left the staging buffer
        // ---- The records of a tile wait in the stage — 16 bits each (position << 2 | flags < 2^16: P.stage_u16, every tiling of
        // up to eight chunks) — and leave ONCE, at the tile's end: a store in the chunk loop sits in the memory queue ahead of the
        // chunk loads that are waited for next (vmcnt counts in order), so a mid-tile flush is a stall; with 16-bit entries the stage
        // holds a tile of random sequence whole (592 records at configs[1]'s geometry; a tile has 350 +- 19).  A tile with more
        // (a telomere: a record every k bases) flushes from inside a pass when the stage is full, as before.
        //
        // What the tile hands to block calling and to a shard's message (EMIT): its VISIBLE records — the canonical ones, and
        // every record where the tile lies in its segment's terminal zone: what a writer reads (src/teloscope.cpp:486-496) — and
        // the summary of its chains of matches (TsTileChain) that lets the interstitial search skip the tile without reading its
        // records.  Both are taken from the records AS THEY LEAVE at the tile's end (whole rows of 64 owned records in tile order, a
        // record per lane), not in the per-match pass: the pass and the chunk loop are then the plain build's — with the emit
        // state live through them the register allocator spilled 43 more scalars there (round 4: +17 % on configs[1]) — and all the
        // state this needs lives for those few rows only.  A tile that flushed from inside a pass (marked: park[5]) is looked at
        // from the wave's region of the output, which holds all its records by then.
        // A record more than -k behind the record before it is a head (a first record more than -k behind the tile's first base
        // is one whatever lies ahead of the tile: the latest position starts at 0; a closer one is not a head HERE — whether it
        // opens a chain depends on the tile before, which the screening looks up: ts_chain_screen, blockcall.hip).
        auto stage_at = [&](uint32_t i) -> uint32_t { return st16 ? (uint32_t)((lds_u16 *)stage)[i] : stage[i]; };
        auto flush_stage = [&]() {                                 // stage[0 .. done - flushed) -> wave_out[cursor + flushed ..), from inside a pass
            __builtin_amdgcn_wave_barrier();
            const uint32_t n = done - flushed;
            KernArgs Q = tail_params();
            const uint32_t cap = Q->region_cap;
            uint32_t *const wave_out = Q->matches_out + (u64)gw * cap;
            const bool rec16 = Q->rec16 != 0u;                       // (16-bit records: the stage's entries as they are, see TsScanParams)
            for (uint32_t i = lane; i < n; i += 64u) {
                const uint32_t o = cursor + flushed + i;
                if (o < cap && !(TS_ABL & 1)) {
                    if (rec16) gstore_lo16((uint16_t *)Q->matches_out + (u64)gw * cap + o, stage_at(i));
                    else gstore(wave_out + o, stage_at(i));
                }
            }
            if (EMIT) park[5] = 1u;                                // (every lane writes the same word)
            // (the stores are waited for here, where a dense tile pays for it, so that the compiler knows of no pending store whose
            // registers the code after the chunk loop reuses: it protected those with a vmcnt(0) at the start of finish_records, which
            // was also a wait for the next tile's first chunk and the ticket's atomic — a memory round trip per tile, dense or not)
            __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
            __builtin_amdgcn_wave_barrier();
            flushed = done;
        };
        // the tile's end: what is left in the stage leaves, and (EMIT) every record of the tile is looked at on its way
        uint32_t tile_vis = 0;                                     // EMIT: visible records of the tile (set by finish_records)
        auto finish_records = [&]() {
            __builtin_amdgcn_wave_barrier();
            KernArgs Q = tail_params();
            const uint32_t cap = Q->region_cap;
            uint32_t *const wave_out = Q->matches_out + (u64)gw * cap;
            // (a lane id the compiler cannot see through: what the rows derive from it — LDS and output addresses, masks — is
            // then computed here, per tile, instead of being kept in registers through the chunk loop, which has none to spare)
            uint32_t ln = lane;
            asm volatile("" : "+v"(ln));
            uint32_t n = done - flushed;
            bool redo = false;
            if (EMIT) redo = __builtin_amdgcn_readfirstlane((int)park[5]) != 0;
            const bool rec16 = Q->rec16 != 0u;
            uint16_t *const wave_out16 = (uint16_t *)Q->matches_out + (u64)gw * cap;
            if (!EMIT || redo) {
                if (!EMIT && st16 && rec16 && cursor + flushed + n <= cap) {
                    // the staged entries ARE the records: copied four to a lane (an 8-byte LDS read, an 8-byte store — unaligned where
                    // the region's cursor is) instead of one; the last one to three go singly
                    const uint32_t n4 = n & ~3u;
                    uint16_t *const dst = wave_out16 + cursor + flushed;
                    for (uint32_t i = 4u * ln; i < n4; i += 256u) {
                        const u32x2 q = *(const LDS u32x2 *)((lds_u16 *)stage + i);
                        if (!(TS_ABL & 1)) gstore_unaligned((unsigned char *)(dst + i), (u64)q.x | ((u64)q.y << 32));
                    }
                    if (ln < n - n4 && !(TS_ABL & 1)) gstore_lo16(dst + n4 + ln, stage_at(n4 + ln));
                } else
                for (uint32_t i = ln; i < n; i += 64u) {
                    const uint32_t o = cursor + flushed + i;
                    if (o < cap && !(TS_ABL & 1)) {
                        if (rec16) gstore_lo16(wave_out16 + o, stage_at(i));
                        else gstore(wave_out + o, stage_at(i));
                    }
                }
                flushed = done;
                if (!EMIT) return;
                // the whole tile again, from the region.  The loads below must find the stores above: same wave, same L2 — the
                // stores have been acknowledged (vmcnt) before the loads are issued, and the loads bypass the CU's non-coherent L1
                // (agent-scope atomic loads).
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                park[5] = 0u;
                n = done;
            }
            const uint32_t base = redo ? 0u : flushed;             // index (in the tile) of row 0's first record
            uint32_t ch_first = 0, ch_last = 0, ch_cc = 0, ch_w1 = 0, vout = 0;
            const uint32_t vbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)park[4]);   // the wave's visible cursor
            // (a scalar load — the table is written by the host only — not a vector one, whose wait (vmcnt) is also a wait for the next
            // tile's first chunk, in flight since the end of phase 1, and for every store ahead of it)
            // P.emit == 2 (a read batch: tips-only, every segment terminal zone as a whole): what leaves is the INDEX, among the
            // tile's records, of every canonical record — the read predicate (predicate.hip: ts_read_predicate_canon) then looks
            // only at the chains that hold one, a twentieth of the records — and no chain summary.
            const bool idx_mode = TS_READ_INDEX_BUILD ? Q->emit == 2u : false;     // (a build flag: two vector instructions per row otherwise)
            typedef const uint32_t __attribute__((address_space(4))) *ConstU32;
            const uint32_t zone = idx_mode ? TS_ZONE_NONE : ((ConstU32)(uintptr_t)Q->tile_zone)[tile];
            const uint32_t vis_cap = Q->vis_cap;
            const bool wide = FAST ? false : Q->vis_wide != 0u;        // (16-bit stage entries = tile positions below 2^14 = 16-bit visible records)
            void *const vis_out = Q->vis_out;
            const u64 vwave = (u64)gw * vis_cap;
            const uint32_t kdist = P.kdist;
            // what a row of up to 64 records (a record per lane, r = 0 in the lanes behind it) adds to the tile's visible records and
            // to its chain summary
            // (whole-tile bounds, checked once on the scalar unit: the per-lane checks are for the tile that does not fit)
            const bool rec_fit = cursor + base + n <= cap, vis_fit = vbase + n <= vis_cap;
            auto look = [&](const uint32_t r, const uint32_t nrow, const uint32_t i0, auto full_row) {
                constexpr bool FULL = decltype(full_row)::value;           // a row of 64 records: no mask of live lanes to make or to apply
                const uint32_t u = r >> 2;                                 // position in the tile
                const u64 live_m = FULL ? ~0ull : low_bits(nrow);
                const u64 canm = ballot64((r & 1u) != 0u);                  // (the lanes behind the row hold 0)
                if (i0 == 0u) ch_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);   // the tile's first record
                // ---- visible records: the canonical ones, and every record where the tile lies in the terminal zone
                {
                    u64 vm = canm;
                    if (zone != TS_ZONE_NONE) vm = live_m & (canm | ballot64(u < (zone & 0xFFFFu)) | ballot64(u >= (zone >> 16)));
                    if (vm != 0ull && !(TS_EMIT_ABL & 2)) {
                        const uint32_t at = vbase + vout + __builtin_amdgcn_mbcnt_hi((uint32_t)(vm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vm, 0u));
                        if (__builtin_amdgcn_inverse_ballot_w64(vm) && (vis_fit || at < vis_cap)) {
                            const uint32_t what = idx_mode ? base + i0 + ln : r;
                            if (wide) gstore((uint32_t *)vis_out + (vwave + at), what);
                            else gstore((uint16_t *)vis_out + (vwave + at), (uint16_t)what);
                        }
                        vout += (uint32_t)__popcll(vm);
                    }
                }
                // ---- chains.  All a row keeps is the canonical count of the chain that is open at its end (ch_cc) and, until
                // the tile's first head, the count ahead of it.  A row without a canonical record and with none carried in (a
                // third of them) changes neither.
                const uint32_t ncan = (uint32_t)__popcll(canm);
                const uint32_t t = ch_cc + ncan;
                if ((t | (~ch_w1 & TS_CHAIN_HEADS)) != 0u && !idx_mode && !(TS_EMIT_ABL & 4)) {            // (integer logic: a uniform bool costs three scalar instructions to combine)
                    // the lane below holds the record before, lane 0 gets the last record of the row before
                    uint32_t below = (uint32_t)__builtin_amdgcn_update_dpp((int)ch_last, (int)u, 0x138, 0xf, 0xf, false);   // wave_shr:1, lane 0 keeps ch_last
                    asm volatile("" : "+v"(below));            // (kept a v_mov_b32_dpp: see lane_below)
                    const u64 H = FULL ? ballot64(u - below > kdist) : live_m & ballot64(u - below > kdist);
                    if (__builtin_expect(t >= 4u, 0)) {
                        // a chain that ends in this row may hold the four canonical records a block needs: the exact look
                        if (H != 0ull) {
                            const u64 ahead = ~H & (H - 1ull);                      // the lanes below the first head
                            const uint32_t top_head = 63u - (uint32_t)__builtin_clzll(H);
                            const u64 below_top = low_bits(top_head);
                            // the chain carried into the row ends at the first head
                            const uint32_t carry = ch_cc + (uint32_t)__popcll(canm & ahead);
                            if (!(ch_w1 & TS_CHAIN_HEADS)) ch_w1 |= carry < 0x7FFFu ? carry : 0x7FFFu;
                            else if (carry >= 4u) ch_w1 |= TS_CHAIN_INNER;
                            // chains that start and end inside the row: looked at only when they hold four canonical records between them
                            if (__popcll(canm & ~ahead & below_top) >= 4) {
                                const u64 above = ln < 63u ? H >> (ln + 1u) : 0ull;
                                const uint32_t next = above ? ln + 1u + (uint32_t)__builtin_ctzll(above) : 64u;   // the next head's lane
                                const bool mine4 = __popcll(canm & low_bits(next) & ~low_bits(ln)) >= 4;
                                if (ballot64(((H >> ln) & 1ull) && ln != top_head && mine4) != 0ull) ch_w1 |= TS_CHAIN_INNER;
                            }
                            ch_w1 |= TS_CHAIN_HEADS;
                            ch_cc = (uint32_t)__popcll(canm & ~below_top);
                        } else {
                            ch_cc = t;
                        }
                    } else {
                        // Fewer than four (nearly every row): nothing to saturate, no chain to flag.  Until the tile's first
                        // head: the canonical count ahead of it, ch_cc + popcount(canm below H's lowest bit), and the HEADS flag,
                        // into ch_w1.  Always: ch_cc = canonical records from H's highest bit on, or t when the row has no head.
                        // Written out for the scalar unit (the compiler's version of these seven lines came to forty
                        // instructions, every combination of two uniform conditions materialised as a 64-bit mask):
                        uint32_t tmp;
                        u64 m;
                        asm volatile("s_bitcmp1_b32 %[w1], 15\n\t"
                                     "s_cbranch_scc1 .Lts_cq%=\n\t"
                                     "s_cmp_eq_u64 %[H], 0\n\t"
                                     "s_cbranch_scc1 .Lts_cq%=\n\t"
                                     "s_ff1_i32_b64 %[tmp], %[H]\n\t"
                                     "s_lshl_b64 %[m], -1, %[tmp]\n\t"
                                     "s_andn2_b64 %[m], %[canm], %[m]\n\t"
                                     "s_bcnt1_i32_b64 %[tmp], %[m]\n\t"
                                     "s_add_i32 %[tmp], %[tmp], %[cc]\n\t"
                                     "s_or_b32 %[w1], %[w1], %[tmp]\n\t"
                                     "s_bitset1_b32 %[w1], 15\n"
                                     ".Lts_cq%=:\n\t"
                                     "s_flbit_i32_b64 %[tmp], %[H]\n\t"          // (-1 without a head: the shift below is then 0 and t is taken anyway)
                                     "s_xor_b32 %[tmp], %[tmp], 63\n\t"
                                     "s_lshl_b64 %[m], -1, %[tmp]\n\t"
                                     "s_and_b64 %[m], %[m], %[canm]\n\t"
                                     "s_bcnt1_i32_b64 %[tmp], %[m]\n\t"
                                     "s_cmp_lg_u64 %[H], 0\n\t"
                                     "s_cselect_b32 %[cc], %[tmp], %[t]"
                                     : [cc] "+s"(ch_cc), [w1] "+s"(ch_w1), [tmp] "=&s"(tmp), [m] "=&s"(m)
                                     : [H] "s"(H), [canm] "s"(canm), [t] "s"(t)
                                     : "scc");
                    }
                }
                ch_last = (uint32_t)__builtin_amdgcn_readlane((int)u, FULL ? 63 : (int)nrow - 1);
            };
            // Two loops, not one with the source of a row picked inside it: a loop that may load from global memory waits for its
            // loads with vmcnt(0), which also waits for the row stores before them and for the next tile's first chunk (in flight
            // since the end of phase 1) — six memory round trips per tile in the first version of this code.
            if (!redo) {
                const uint32_t nfull = TS_EMIT_FULL_ROWS ? n & ~63u : 0u;
                for (uint32_t i0 = 0; i0 < nfull; i0 += 64u) {             // whole rows
                    const uint32_t o = cursor + base + i0 + ln;
                    const uint32_t r = stage_at(i0 + ln);
                    if (!(TS_ABL & 1)) {
                        auto put = [&]() { if (rec16) gstore_lo16(wave_out16 + o, r); else gstore(wave_out + o, r); };
                        if (rec_fit) put();                                // (whole-tile bound, scalar)
                        else if (o < cap) put();
                    }
                    if (!(TS_EMIT_ABL & 1)) look(r, 64u, i0, std::true_type{});
                }
                for (uint32_t i0 = nfull; i0 < n; i0 += 64u) {              // the tile's last, partial row
                    const uint32_t nrow = n - i0 < 64u ? n - i0 : 64u;
                    const uint32_t o = cursor + base + i0 + ln;
                    uint32_t r = 0u;
                    if (ln < nrow) {
                        r = stage_at(i0 + ln);
                        if (o < cap && !(TS_ABL & 1)) {
                            if (rec16) gstore_lo16(wave_out16 + o, r);
                            else gstore(wave_out + o, r);
                        }
                    }
                    if (!(TS_EMIT_ABL & 1)) look(r, nrow, i0, std::false_type{});
                }
            } else {
                for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
                    const uint32_t nrow = n - i0 < 64u ? n - i0 : 64u;
                    const uint32_t o = cursor + base + i0 + ln;
                    uint32_t r = 0u;
                    if (ln < nrow && o < cap)                             // (past the non-coherent L1)
                        r = rec16 ? (uint32_t)__hip_atomic_load(wave_out16 + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : __hip_atomic_load(wave_out + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    look(r, nrow, i0, std::false_type{});
                }
            }
            flushed = done;
            // the tile's summary and where its visible records are
            const uint32_t z15 = ch_cc < 0x7FFFu ? ch_cc : 0x7FFFu;
            if (!(ch_w1 & TS_CHAIN_HEADS)) ch_w1 |= z15;               // no head: every canonical record is ahead of the first
            const u64 voff = vwave + vbase;
            if (ln == 0u) gstore((uint4 *)&Q->tile_chain[4ull * tile], make_uint4((ch_first & 0xFFFFu) | (ch_last << 16), ch_w1 | (z15 << 16), (uint32_t)voff, (uint32_t)(voff >> 32)));
            tile_vis = vout;
            park[4] = vbase + vout;                                // (every lane writes the same word)
        };
        if (P.windows_on)                                          // match fields accumulate from zero
            for (uint32_t it = lane; it < acc_rows * 2u * (P.acc_copies + (accb ? 1u : 0u)); it += 64u) ((lds_u32 *)wacc)[it] = 0u;

        // ------------------------------------------------------------------ phase 1
        // Chunk c+1's 16 B/lane load is in flight while chunk c is resolved; the loop is unrolled
        // by two with alternating registers so the loaded value is never copied (a copy would
        // make the compiler wait for the load it was meant to overlap).
        // ---- per-match work on full wavefronts.  Matches are a few per cent of the positions, so a
        // lane-per-position loop would idle most lanes: instead every chunk appends its match positions,
        // compacted in position order (prefix sum over the wave, then each lane writes out its set bits),
        // to a queue in LDS, and the queue is consumed 64 matches at a time: flags from the flag table,
        // the packed record, and the match's contribution to every window that contains it.  A pass only
        // runs on 64 queued matches (fewer when the tile ends or the queue must make room).
        uint32_t qhead = 0, qcount = 0;                            // head as a byte offset into the ring, entries queued
        auto drain_queue = [&](const uint32_t threshold) {
            if (TS_ABL & 128) { qcount = 0; return; }           // profiling: compaction only, nothing consumed
            if (qcount >= threshold && qcount > 0u) set_prio(kPrioPass);
            while (qcount >= threshold && qcount > 0u) {
                const uint32_t n = qcount < 64u ? qcount : 64u;
                if (done - flushed + 64u > P.stage_cap) flush_stage();
                const bool live = lane < n;
                const uint32_t xp = live ? (uint32_t)*lds_at16(((qhead + lane * 2u) & qmask) | qbase) : 0u;   // plane coord of the match
                // its k-mer, from the code plane (16 positions per dword)
                const lds_u32 *cw = codes + __builtin_amdgcn_ubfe(xp, 4, 12);
                const uint32_t kw = __builtin_amdgcn_alignbit(cw[1], cw[0], (xp & 15u) * 2u);
                const uint32_t idx = kw & kmask;
                uint32_t fc;                                  // forward << 1 | canonical
                if (PAIR_BYTES) fc = __builtin_amdgcn_ubfe((uint32_t)*lds_at8(kw & pmask), 2, 2);   // bits 2..3 of the pair-table byte (table at LDS address 0)
                else if (FC_BYTES) fc = fc_bytes[idx];
                else fc = (fc_table[idx >> 4] >> ((idx & 15u) << 1)) & 3u;
                // position in the tile, step block and offset inside it (a tips-only tile is one block: nothing to divide).  u < 2^16,
                // so the quotient is one 24-bit multiply (full rate): by ceil(2^22 / s) where that is exact for every position of a
                // tile (P.div_exact: the host checked (nch x 2016 + 64) x (ceil(2^22 / s) s - 2^22) < 2^22 — s = 500, 1000, ...), else
                // by ceil(2^16 / s), exact or one too large and corrected
                const uint32_t u = xp - sh;                   // wraps for the few bases before the tile
                uint32_t q = 0u, o = u;
                if (P.windows_on) {
                    uint32_t qs;
                    if (P.div_exact) {
                        q = __umul24(u, P.s_magic22) >> 22; qs = __umul24(q, P.s);
                    } else {
                        q = __umul24(u, P.s_inv) >> 16; qs = __umul24(q, P.s);
                        if (qs > u) { --q; qs -= P.s; }
                    }
                    o = u - qs;
                }
                // Which lanes hold what is decided on MASKS — one vector compare each, the logic between them on the scalar unit —
                // and a lane's own predicate is read back off the mask (inverse ballot: the mask itself becomes the lane predicate).
                // w == s: a match that would straddle a window end is lost (the carry rule of
                // src/teloscope.cpp:611-628; pinned by t2t.fa -i = 199)
                u64 valid_m = low_bits(n) & ballot64(xp >= sh);
                if (P.straddle_fix) valid_m &= ~ballot64(o + k > P.s);
                const u64 bal = valid_m & ballot64(xp < own_end);          // the owned matches of this pass
                const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_m), owned = __builtin_amdgcn_inverse_ballot_w64(bal);
                // record slot = rank among the owned matches of this pass
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                const uint32_t slot = owned ? (done - flushed) + rank : P.stage_cap + lane;
                const uint32_t record = (u << 2) | fc;
                if (st16) ((lds_u16 *)stage)[slot] = (uint16_t)record; else stage[slot] = record;
                done += (uint32_t)__popcll(bal);
                const u64 can_all = ballot64((fc & 1u) != 0u), fwd_all = ballot64((fc & 2u) != 0u);
                const bool is_can = __builtin_amdgcn_inverse_ballot_w64(can_all), is_fwd = __builtin_amdgcn_inverse_ballot_w64(fwd_all);
                const u64 canm = can_all & bal;
                ccan += (uint32_t)__popcll(canm);              // (scalar counts: popcounts of masks)
                cfwd += (uint32_t)__popcll(fwd_all & bal);
                // {canonical, non-canonical, forward, reverse} as one 4 x 16-bit increment
                const u64 inc = (u64)(is_can ? 1u : 0x10000u) | ((u64)(is_fwd ? 1u : 0x10000u) << 32);
                // windows q, q-1, ... contain the match as long as it ends inside them.  Window q always does:
                // o + k <= w holds for every valid match (w == s: the straddle rule above; w > s: k <= w - s)
                if (P.windows_on && accb && !(TS_ABL & 4)) {
                    // per step block: block q takes the match, and the row of run-overs takes it too when it ends behind the block's
                    // end (o + k > s: k - 1 offsets of s) — a window is the sum of its blocks minus the run-overs of its last one
                    if (valid && q < nblk) atomicAdd((unsigned long long *)(wacc + acc_off + q), inc);
                    const u64 over = valid_m & ballot64(o + k > P.s);
                    if (over != 0ull) {
                        if (__builtin_amdgcn_inverse_ballot_w64(over) && q < nblk) atomicAdd((unsigned long long *)(wacc + P.acc_copies * acc_rows + q), inc);
                    }
                } else if (P.windows_on && !(TS_ABL & 4)) {
                    for (uint32_t j = 0; j < nwper; ++j) {
                        const uint32_t wi = q - j;            // wraps past window 0
                        const bool in = valid && wi < T.nwin && o + k + j * P.s <= P.w;
                        if (in) atomicAdd((unsigned long long *)(wacc + acc_off + wi), inc);
                    }
                }
                qhead = (qhead + 2u * n) & (TS_LIST * 2u - 1u);
                qcount -= n;
            }
            set_prio(0);
            __builtin_amdgcn_wave_barrier();                  // the queue is appended to next
        };

        auto scan_chunk = [&](const uint32_t cpos, const uint32_t ch, const uint4 v0, const uint4 v1) -> uint32_t {
            // cpos = c * TS_CHUNK, ch = c * 63.  A lane holds 32 consecutive bases (two packed dwords).
            // ASCII -> 2-bit codes (A0 C1 T2 G3) and a validity check, 4 bases per dword
            const uint32_t x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            // ASCII & 6 = twice the code (A 0, C 2, T 4, G 6) and at the same time the v_perm selector of the
            // letter that code stands for (8-byte LUT, letters at the even bytes); any byte that is not that
            // letter (up to case folding) leaves a bit in `bad`: three instructions per four bases
            uint32_t t[8], e[8], bad = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                t[i] = x[i] & 0x06060606u;
                e[i] = __builtin_amdgcn_perm(0xFF47FF54u, 0xFF43FF41u, t[i]);
                bad = __builtin_amdgcn_bitop3_b32(x[i], e[i], bad, 0xBE);      // bad | (x ^ e), one v_bitop3_b32
            }
            uint32_t wa = pack16(t), wb = pack16(t + 4);

            const uint32_t pos0 = cpos + lane * 32u;             // plane coord of this lane's first base
            uint32_t inv = 0;                                     // invalid bases among the lane's 32
            const bool slow = __any((bad & P.fold_mask) != 0u) || (cpos + 2048u > xend);
            if (slow) {                                           // wave-uniform, rare
                asm volatile("; non-ACGT bases or segment end in this chunk" ::: "memory");   // keeps the compiler from hoisting this block
                uint32_t b4[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t d = (x[i] & P.fold_mask) ^ e[i];
                    const uint32_t nz = ((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u) >> 7;
                    b4[i] = __builtin_amdgcn_udot4(nz, 0x08040201u, 0u, false);
                }
                inv = b4[0] | (b4[1] << 4) | (b4[2] << 8) | (b4[3] << 12) |
                      (b4[4] << 16) | (b4[5] << 20) | (b4[6] << 24) | (b4[7] << 28);
                // non-ACGT bytes among the tile's own positions [sh, xend): the window phase then counts the tile's
                // nucleotides from the bytes, not from the codes (what lies before the tile or behind the segment
                // end is never counted)
                uint32_t inside = inv;
                if (pos0 < sh) inside &= ~0u << (sh - pos0);             // chunk 0, lane 0: sh < 16
                if (pos0 + 32u > xend) {
                    const uint32_t nv = xend > pos0 ? xend - pos0 : 0u;     // < 32
                    inside &= ~(~0u << nv);
                    inv |= ~0u << nv;
                }
                if (__any(inside != 0u)) has_invalid = true;
                wa = pack16(t); wb = pack16(t + 4);
            }

            // next lane's first 16 bases (DPP wave_shl:1, lane i <- lane i+1; lane 63's value is unused)
            const uint32_t nx = (uint32_t)__builtin_amdgcn_mov_dpp((int)wa, 0x130, 0xf, 0xf, false);

            // ONE table probe per TWO positions: the pair table is indexed by the (k+1)-mer that
            // starts at an even position and holds {match at p, match at p+1} (16 entries per dword:
            // row = index >> 4, bit = 2 * (index & 15)).  All sixteen ds_read_b32 are issued back to
            // back (inline asm) and consumed behind one counted wait.
#if !(TS_ABL & 32)
            uint32_t tmp[16], ent[16];
            // bases 16..31 of a code dword followed by bases 0..15 of the next one: the (k+1)-mers that start
            // at bases 10, 12, 14 lie inside it (k <= 6)
            const uint32_t xa = PAIR_BYTES ? __builtin_amdgcn_alignbit(wb, wa, 16) : 0u;
            const uint32_t xb = PAIR_BYTES ? __builtin_amdgcn_alignbit(nx, wb, 16) : 0u;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (PAIR_BYTES) {
                    // one byte per (k+1)-mer: index = address; every probe is a single v_bfe
                    tmp[j] = 0;
                    const uint32_t word = (j & 7) <= 4 ? (j < 8 ? wa : wb) : (j < 8 ? xa : xb);
                    const uint32_t bit = (j & 7) <= 4 ? 4 * (j & 7) : 4 * ((j & 7) - 4);
                    const uint32_t addr = __builtin_amdgcn_ubfe(word, bit, 2u * (k + 1u)) | tab_base;
                    asm volatile("ds_read_u8 %0, %1" : "=v"(ent[j]) : "v"(addr));
                    continue;
                }
                if (j < 8) tmp[j] = (j == 0) ? wa : __builtin_amdgcn_alignbit(wb, wa, 4 * j);
                else tmp[j] = (j == 8) ? wb : __builtin_amdgcn_alignbit(nx, wb, 4 * (j - 8));
                {                                                 // 16 entries per dword: row, then shift
                    const uint32_t row = __builtin_amdgcn_ubfe(tmp[j], 4, rowbits);
                    const uint32_t addr = tab_base + (row << 2);
                    asm volatile("ds_read_b32 %0, %1" : "=v"(ent[j]) : "v"(addr));
                }
            }
#endif
            // (issued while the probes are in flight: the stores and the counting need nothing from them)
            // the lane's bases go to the tile's code plane (lane 63's first dword is the look-ahead of lane 62's
            // last k-mers; the next chunk's lane 0 rewrites the same slot with the same value)
            if (!(TS_ABL & 64)) {
                const uint32_t h = ch + lane;                     // index of the lane's 32 positions in the plane
                *(LDS u32x2 *)(codes + 2u * h) = (u32x2){wa, wb};
            }

#if TS_ABL & 32
            uint32_t M32 = wa & wb & (wa >> 7) & (nx >> 3) & (wb >> 11);
#else
            uint32_t M32 = 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                M32 = __builtin_amdgcn_alignbit(PAIR_BYTES ? ent[j] : ent[j] >> ((tmp[j] << 1) & 31u), M32, 2);
#endif

            if (slow) {                                           // k-mers touching an invalid base
                const uint32_t inv_next = (uint32_t)__builtin_amdgcn_mov_dpp((int)inv, 0x130, 0xf, 0xf, false);
                const u64 iv = (u64)inv | ((u64)inv_next << 32);
                uint32_t kb = 0;
                for (uint32_t i = 0; i < k; ++i) kb |= (uint32_t)(iv >> i);
                M32 &= ~kb;
            }

#if TS_ABL & 16
            M32 = 0;
#endif
            if (lane == 63u) M32 = 0;                             // lane 63 only looks ahead for lane 62
            return M32;
        };
        // One chunk's matches onto the queue (nm = the lane's count, incl = its inclusive prefix sum, total).
        // A chunk with more matches than the queue holds (dense repeats) is appended one lane group at a time —
        // a group's matches always fit, and groups are in position order.
        auto append_chunk = [&](const uint32_t M32, const uint32_t cpos, const uint32_t nm0, const uint32_t incl0, const uint32_t total0) {
            const uint32_t lbase = cpos + lane * 32u;             // plane coordinates, < 65536 (nch <= 32)
            constexpr uint32_t kGroup = TS_LIST / 32u;
            const bool dense = TS_LIST < TS_CHUNK && total0 > TS_LIST;
            const uint32_t ngroups = dense ? 64u / kGroup : 1u;
            for (uint32_t gi = 0; gi < ngroups; ++gi) {
                uint32_t m = M32, nm = nm0, incl = incl0, total = total0;
                if (dense) {
                    m = (lane / kGroup == gi) ? M32 : 0u;
                    nm = __popc(m);
                    incl = wave_scan_incl(nm);
                    total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
                // the queue keeps whatever is short of a full pass; it is emptied first if these
                // matches would not fit behind it
                drain_queue(qcount + total > TS_LIST ? 1u : 64u);
                uint32_t o = qhead + 2u * (qcount + incl - nm);                   // byte offset of the lane's first slot
                while (m) {
                    *lds_at16((o & qmask) | qbase) = (uint16_t)(lbase + (uint32_t)__builtin_ctz(m));
                    o += 2u;
                    m &= m - 1u;
                }
                qcount += total;
                __builtin_amdgcn_wave_barrier();
            }
        };
        // The matches of TWO consecutive chunks go onto the queue behind one prefix sum (both per-lane counts
        // ride in one register), one drain decision and one pass over the queue: A's matches first, then B's.
        auto append_pair = [&](const uint32_t mA, const uint32_t cposA, const uint32_t mB, const uint32_t cposB) {
            if (!__any((mA | mB) != 0u)) return;
            set_prio(kPrioAppend);
            const uint32_t nA = __popc(mA), nB = __popc(mB);
            const uint32_t incl = wave_scan_incl(nA | (nB << 16));
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t totA = tot & 0xFFFFu, totB = tot >> 16;
            if (totA + totB + 63u > TS_LIST) {                    // rare: does not fit behind what a pass may leave
                asm volatile("; dense pair of chunks" ::: "memory");
                if (totA) append_chunk(mA, cposA, nA, incl & 0xFFFFu, totA);
                if (totB) append_chunk(mB, cposB, nB, incl >> 16, totB);
                set_prio(0);
                return;
            }
            drain_queue(64u);                                     // leaves fewer than 64 queued
            set_prio(kPrioAppend);
            uint32_t o = qhead + 2u * (qcount + (incl & 0xFFFFu) - nA), m = mA;
            uint32_t lbase = cposA + lane * 32u;
            while (m) {
                *lds_at16((o & qmask) | qbase) = (uint16_t)(lbase + (uint32_t)__builtin_ctz(m));
                o += 2u;
                m &= m - 1u;
            }
            o = qhead + 2u * (qcount + totA + (incl >> 16) - nB); m = mB;
            lbase = cposB + lane * 32u;
            while (m) {
                *lds_at16((o & qmask) | qbase) = (uint16_t)(lbase + (uint32_t)__builtin_ctz(m));
                o += 2u;
                m &= m - 1u;
            }
            qcount += totA + totB;
            set_prio(0);
            __builtin_amdgcn_wave_barrier();
        };
        {
            // Chunk c+1's two 16 B/lane loads are in flight while chunk c is resolved; the loop is
            // unrolled by two with alternating registers so the loaded values are never copied, and
            // loads are unconditional (the last ones re-read the final chunk) so that the wait before
            // each resolve is a counted vmcnt(2), never vmcnt(0).  Chunk offsets are carried as
            // scalars that step by a constant: no vector multiplies.
            const uint32_t last_pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)((nch - 1u) * TS_CHUNK));
            const unsigned char *lsrc = src + lane * 32u;
            uint4 a0 = n0, a1 = n1, b0, b1;
            uint32_t cpos = 0, ch = 0;
            for (uint32_t c = 0; c < nch; c += 2u) {
                const uint32_t p1 = cpos + TS_CHUNK < last_pos ? cpos + TS_CHUNK : last_pos;
                b0 = *(const uint4 *)(lsrc + p1);
                b1 = *(const uint4 *)(lsrc + p1 + 16);
                __builtin_amdgcn_sched_barrier(0);         // the loads are issued before chunk c is touched
                const uint32_t mA = scan_chunk(cpos, ch, a0, a1);
                uint32_t mB = 0u;
                const bool pair = c + 1u < nch;
                if (pair) {
                    const uint32_t p2 = cpos + 2u * TS_CHUNK < last_pos ? cpos + 2u * TS_CHUNK : last_pos;
                    a0 = *(const uint4 *)(lsrc + p2);
                    a1 = *(const uint4 *)(lsrc + p2 + 16);
                    __builtin_amdgcn_sched_barrier(0);
                    mB = scan_chunk(cpos + TS_CHUNK, ch + 63u, b0, b1);
                }
                append_pair(mA, cpos, mB, cpos + TS_CHUNK);
                cpos += 2u * TS_CHUNK; ch += 126u;
            }
        }
#define TS_REQUEST_NEXT_TILE()                                                                                                   \
        {                                                                                                                        \
            /* the chunk loop's last loads re-read the final chunk and are never used: told complete here (they are a chunk old),  \
               or every register of theirs that is written below waits for them AND for the loads requested here */               \
            if (TS_ASM_STORES) __builtin_amdgcn_s_waitcnt(0x0F70);                                                                \
            KernArgs Q = tail_params();                                                                                          \
            const bool dyn = Q->dynamic_tiles != 0u;                                                                             \
            tile_next = dyn ? group + Q->ticket_groups * (group_waves + (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket))   \
                            : tile + total_waves;                                                                                \
            if (tile_next < P.ntiles) {           /* next tile: descriptor + first chunk, in flight during phase 2 */            \
                const unsigned char *ls = P.in + (const_tiles[tile_next].in_off & ~15ull) + lane * 32u;                          \
                n0 = *(const uint4 *)ls; n1 = *(const uint4 *)(ls + 16);                                                         \
                if (dyn && lane == 0) ticket = take_ticket(Q);      /* and the ticket for the tile after it */                   \
            }                                                                                                                    \
        }
        if (!(EMIT && TS_EMIT_LATE_REQUEST)) TS_REQUEST_NEXT_TILE()
        drain_queue(1u);                          // the matches still queued when the tile ends
        __builtin_amdgcn_wave_barrier();          // planes written above are read by other lanes below

        // ------------------------------------------------------------------ phase 2: windows
        set_prio(kPrioWindows);
        // The match fields of the tile's window records are complete (accumulated above).
        // (the emitting build flushes — and looks at — the records BEHIND the window phase: its rows want thirty scalars of their own,
        // and between the chunk loop and the window phase they competed with both for registers: 110 more spill moves per tile)
        if (EMIT ? !TS_EMIT_FINISH_LAST : !TS_PLAIN_FINISH_LAST) finish_records();
        if (EMIT && TS_EMIT_LATE_REQUEST) TS_REQUEST_NEXT_TILE()
        if (P.windows_on) {
            // Nucleotides.  Counted here, from the tile's code plane, not per chunk: a row (a step block when w is a
            // multiple of s — a window is then the sum of w / s of them and overlapping windows share them — else a
            // window) is cut into pieces for 1, 2 or 4 lanes, and a lane counts its piece 32 positions at a time:
            // the low code bits of two code dwords merged into one dword, the high bits into another (codes A0 C1 T2
            // G3: low bit set in C and G, high bit in T and G), three popcounts — 4 instructions per 16 positions on
            // full wavefronts, against 23 per chunk for per-lane byte planes plus their sums (v5 .. v11).
            auto count_piece = [&](uint32_t xs, uint32_t xe, uint32_t &nL, uint32_t &nH, uint32_t &nG) {   // plane coords, xs < xe
                const uint32_t d0 = xs >> 4, d1 = (xe - 1u) >> 4;            // code dwords of the piece
                const uint32_t m0 = ~0u << (2u * (xs & 15u));
                const uint32_t m1 = ~0u >> (2u * (15u - ((xe - 1u) & 15u)));
                auto add2 = [&](uint32_t wA, uint32_t wB) {                  // two code dwords = 32 positions
                    const uint32_t lo = (wA & 0x55555555u) | ((wB << 1) & 0xAAAAAAAAu);
                    const uint32_t hi = ((wA >> 1) & 0x55555555u) | (wB & 0xAAAAAAAAu);
                    nL += __popc(lo); nH += __popc(hi); nG += __popc(lo & hi);
                };
                // the two end dwords (masked) as one pair, the dwords between them two at a time
                uint32_t first = codes[d0] & m0, last = 0u;
                if (d1 == d0) first &= m1; else last = codes[d1] & m1;
                add2(first, last);
                uint32_t d = d0 + 1u;
                for (; d + 2u <= d1; d += 2u) add2(codes[d], codes[d + 1u]);
                if (d < d1) add2(codes[d], 0u);
            };
            // the same from the bytes themselves, in tiles with non-ACGT bytes (rare: segments end at gaps)
            auto count_piece_bytes = [&](uint32_t xs, uint32_t xe, uint32_t n[4]) {
                const uint32_t fold = P.fold_mask & 0xFFu;
                for (uint32_t x = xs; x < xe; ++x) {
                    const uint32_t c = (uint32_t)src[x] & fold;
                    n[0] += c == 0x41u; n[1] += c == 0x43u; n[2] += c == 0x47u; n[3] += c == 0x54u;
                }
            };
            // When w is a multiple of s rec[] holds one row per step block (the tile's windows + halo), otherwise one
            // per window.
            const bool by_blocks = P.block_sums != 0u;
            if (P.nuc_on && !(TS_ABL & 2)) {
                const uint32_t rows = by_blocks ? T.nwin + P.halo_blocks : T.nwin;
                const uint32_t len = by_blocks ? P.s : P.w;
                for (uint32_t it = lane; it < rows * 4u; it += 64u) rec[it] = 0u;
                // pieces per row: as many lanes as there are busy, as long as a piece keeps 128 positions
                uint32_t pshift = 0;
                while (pshift < 2u && (rows << (pshift + 1u)) <= 64u && (len >> (pshift + 1u)) >= 128u) ++pshift;
                const uint32_t piece = (len + (1u << pshift) - 1u) >> pshift;
                __builtin_amdgcn_wave_barrier();
                for (uint32_t it = lane; it < (rows << pshift); it += 64u) {
                    const uint32_t row = it >> pshift, part = it & ((1u << pshift) - 1u);
                    const uint32_t row_end = row * P.s + len < T.nrel ? row * P.s + len : T.nrel;   // a halo block may lie past the segment end
                    const uint32_t us = row * P.s + part * piece;
                    const uint32_t ue = us + piece < row_end ? us + piece : row_end;
                    if (us >= ue) continue;
                    uint32_t n[4] = {0u, 0u, 0u, 0u};                         // A C G T
                    if (has_invalid) {
                        count_piece_bytes(sh + us, sh + ue, n);
                    } else {
                        uint32_t nL = 0, nH = 0, nG = 0;
                        count_piece(sh + us, sh + ue, nL, nH, nG);
                        n[0] = (ue - us) + nG - nL - nH; n[1] = nL - nG; n[2] = nG; n[3] = nH - nG;
                    }
                    LDS uint32_t *r4 = rec + row * 4u;
                    if (pshift == 0u) { r4[0] = n[0]; r4[1] = n[1]; r4[2] = n[2]; r4[3] = n[3]; }
                    else {
#pragma unroll
                        for (int f = 0; f < 4; ++f) (void)__hip_atomic_fetch_add(r4 + f, n[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();

            // the four packed 16-bit match counters of window i {canonical, non-canonical, forward, reverse}.  The counters of the
            // copies (and of a window's blocks) add without carries between fields: a field's total is at most the matches of one
            // window (<= 32768); the run-overs of the window's last block are among what was added, so the subtraction borrows nothing
            auto window_fields = [&](uint32_t i) -> u64 {
                u64 a = 0;
                if (accb) {
                    for (uint32_t c = 0; c < P.acc_copies; ++c)
                        for (uint32_t j = 0; j < nwper; ++j) a += wacc[c * acc_rows + i + j];
                    a -= wacc[P.acc_copies * acc_rows + i + nwper - 1u];
                } else {
                    for (uint32_t c = 0; c < P.acc_copies; ++c) a += wacc[c * acc_rows + i];
                }
                return a;
            };
            // records leave as whole 16-byte halves in order: {A, C, G, T} and {canonical, non-canonical, forward,
            // reverse} covered bases (= k x matches) per window, coalesced
            if (EMIT && tail_params()->win_packed != nullptr) {
                // A shard's scan (ts_batch_bind_shard_message): the records of the OWNED windows go straight into the message's
                // window section in its bit-packed form — fields [A C G T] (when nucleotide counts are on), canonical,
                // non-canonical and forward match COUNTS, win_field_bits each, least significant first, in win_pack_bytes bytes
                // (9 instead of 32 at w = 1000: what ts_shard_pack_windows, shard.hip, made of the 32-byte records in a pass of
                // its own) — and nothing is written for the context tiles' windows, which nobody reads.  A lane per window.
                KernArgs Q = tail_params();
                const u64 T_win_out = const_tiles[tile].win_out;
                if (T_win_out >= Q->win_pack_lo && T_win_out < Q->win_pack_hi && !(TS_ABL & 8)) {
                    const uint32_t wb = Q->win_pack_bytes, B = Q->win_field_bits;
                    unsigned char *const dst0 = Q->win_packed + T_win_out * (u64)wb;
                    uint32_t ln = lane;
                    asm volatile("" : "+v"(ln));              // (nothing derived from the lane id here is worth a register through the chunk loop)
                    const bool ten = B == 10u && P.nuc_on != 0u;   // seven 10-bit fields (512 <= w < 1024): the layout by constants
                    for (uint32_t i = ln; i < T.nwin; i += 64u) {
                        u64 lo = 0, hi = 0;
                        uint32_t at = 0;
                        auto put = [&](uint32_t v) {
                            lo |= at < 64u ? (u64)v << at : 0ull;
                            if (at + B > 64u) hi |= at >= 64u ? (u64)v << (at - 64u) : (u64)v >> (64u - at);
                            at += B;
                        };
                        uint4 v = make_uint4(0u, 0u, 0u, 0u);
                        if (P.nuc_on) {
                            const uint32_t nrow = by_blocks ? P.halo_blocks + 1u : 1u;
                            for (uint32_t j = 0; j < nrow; ++j) {
                                const LDS uint32_t *r4 = rec + (i + j) * 4u;
                                v.x += r4[0]; v.y += r4[1]; v.z += r4[2]; v.w += r4[3];
                            }
                        }
                        const u64 a = window_fields(i);
                        const uint32_t f4 = (uint32_t)a & 0xFFFFu, f5 = (uint32_t)a >> 16, f6 = (uint32_t)(a >> 32) & 0xFFFFu;
                        if (ten) {
                            // fields at bits 0, 10, .. 60 of a 70-bit record (every field < 1024): ten 32-bit instructions where the
                            // generic shifts by a run-time amount below came to eighty
                            const uint32_t w0 = v.x | (v.y << 10) | (v.z << 20) | (v.w << 30);
                            const uint32_t w1 = (v.w >> 2) | (f4 << 8) | (f5 << 18) | (f6 << 28);
                            lo = (u64)w0 | ((u64)w1 << 32);
                            hi = (u64)(f6 >> 4);
                        } else {
                            if (P.nuc_on) { put(v.x); put(v.y); put(v.z); put(v.w); }
                            put(f4); put(f5); put(f6);
                        }
                        // (unaligned 8- and 4-byte stores: two store instructions for the 9 bytes of w = 1000, not nine)
                        unsigned char *d = dst0 + (u64)i * wb;
                        uint32_t b = 0;
                        if (wb >= 8u) { gstore_unaligned(d, lo); b = 8u; }
                        if (wb - b >= 4u) { const uint32_t x = b ? (uint32_t)hi : (uint32_t)lo; gstore_unaligned(d + b, x); b += 4u; }
                        for (; b < wb; ++b) gstore(d + b, (unsigned char)(b < 8u ? lo >> (8u * b) : hi >> (8u * (b - 8u))));
                    }
                }
            } else if (!(TS_ABL & 8)) {
                uint4 *wout = (uint4 *)(tail_params()->windows_out + const_tiles[tile].win_out * 8ull);
                for (uint32_t it = lane; it < T.nwin * 2u; it += 64u) {
                    const uint32_t i = it >> 1;
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (!(it & 1u)) {
                        if (P.nuc_on) {
                            const uint32_t nrow = by_blocks ? P.halo_blocks + 1u : 1u;
                            for (uint32_t j = 0; j < nrow; ++j) {
                                const LDS uint32_t *r4 = rec + (i + j) * 4u;
                                v.x += r4[0]; v.y += r4[1]; v.z += r4[2]; v.w += r4[3];
                            }
                        }
                    } else {
                        const u64 a = window_fields(i);
                        v.x = ((uint32_t)a & 0xFFFFu) * k; v.y = ((uint32_t)a >> 16) * k;
                        v.z = ((uint32_t)(a >> 32) & 0xFFFFu) * k; v.w = (uint32_t)(a >> 48) * k;
                    }
                    gstore(wout + it, v);
                }
            }
        }

        if (EMIT ? TS_EMIT_FINISH_LAST : TS_PLAIN_FINISH_LAST) finish_records();
        // ------------------------------------------------------- tile directory
        {
            const uint32_t tcan = ccan, tfwd = cfwd;
            if (lane == 0) {
                KernArgs Q = tail_params();
                gstore((u64 *)&Q->tile_off[tile], (u64)gw * Q->region_cap + cursor);
                const uint32_t vout = EMIT ? tile_vis : 0u;       // (the tile's chain summary left with its records: finish_records)
                gstore((uint4 *)&Q->tile_stats[4ull * tile], make_uint4(done, tcan, tfwd, (TS_EXP & 8) ? (((uint32_t)wall_clock64() & 0xFFFFFu) | (gw << 20)) : vout));
            }
            cursor += done;
        }
        set_prio(0);
        __builtin_amdgcn_wave_barrier();          // next tile overwrites the planes
    }
    if (lane == 0) {
        KernArgs Q = tail_params();
        Q->wave_fill[gw] = cursor;                             // records needed by this wave (may exceed region_cap)
        if (EMIT) Q->wave_fill[total_waves + gw] = park[4];    // visible records needed (may exceed vis_cap)
    }
}

// Per-segment hit summary {windows, matches, canonical, forward}: the buffer ranks gather.
__global__ void ts_segment_summary(const uint32_t *tile_stats, const uint32_t *seg_first_tile,
                                   const uint64_t *seg_nwin, uint32_t nseg, u64 *out) {
    // one workgroup per segment (a 250 Mb contig has ~35 k tiles): strided partial sums, LDS reduce
    __shared__ u64 part[3][256];
    const uint32_t sidx = blockIdx.x;
    if (sidx >= nseg) return;
    const uint32_t t0 = seg_first_tile[sidx], t1 = seg_first_tile[sidx + 1];
    u64 nm = 0, nc = 0, nf = 0;
    for (uint32_t t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
        const uint4 st = *(const uint4 *)&tile_stats[4ull * t];
        nm += st.x; nc += st.y; nf += st.z;
    }
    part[0][threadIdx.x] = nm; part[1][threadIdx.x] = nc; part[2][threadIdx.x] = nf;
    __syncthreads();
    for (uint32_t o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) {
            part[0][threadIdx.x] += part[0][threadIdx.x + o];
            part[1][threadIdx.x] += part[1][threadIdx.x + o];
            part[2][threadIdx.x] += part[2][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[4ull * sidx + 0] = seg_nwin[sidx];
        out[4ull * sidx + 1] = part[0][0];
        out[4ull * sidx + 2] = part[1][0];
        out[4ull * sidx + 3] = part[2][0];
    }
}

// Packs the per-wave record regions into one dense stream (used before a D2H copy).
__global__ void ts_compact_regions(const uint32_t *regions, const uint32_t *wave_fill, const u64 *wave_dense_base,
                                   uint32_t region_cap, uint32_t nwaves, uint32_t *dense, int rec16) {
    const uint32_t w = blockIdx.x;
    if (w >= nwaves) return;
    const uint32_t n = wave_fill[w] < region_cap ? wave_fill[w] : region_cap;
    uint32_t *dst = dense + wave_dense_base[w];
    // (the dense stream is 32 bits per record whatever the regions hold: what leaves the device keeps its format)
    if (rec16) { const uint16_t *src = (const uint16_t *)regions + (u64)w * region_cap; for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i]; }
    else { const uint32_t *src = regions + (u64)w * region_cap; for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i]; }
}

}  // namespace

int ts_k_lds_bytes(const TsScanParams *p) { return (int)lds_total(*p); }

namespace {
// The 80-VGPR build (two workgroups per CU) exists for the byte tables only: the probe path of the 2-bit tables (k >= 7)
// needs 97 registers and would spill to scratch there, so it is not instantiated and plan_geometry never asks for it.
template <bool EMIT>
const void *scan_variant_e(const TsScanParams *p) {
    const bool two = p->wgs_per_cu > 1u;
    if (p->pair_byte_table && p->fc_byte_table) {
        const bool fast = p->stage_u16 && (!p->windows_on || p->acc_blocks);
        if (fast) return two ? (const void *)ts_scan_tiles<true, true, 6, EMIT, true> : (const void *)ts_scan_tiles<true, true, 4, EMIT, true>;
        return two ? (const void *)ts_scan_tiles<true, true, 6, EMIT, false> : (const void *)ts_scan_tiles<true, true, 4, EMIT, false>;
    }
    if (two) return nullptr;
    if (p->fc_byte_table) return (const void *)ts_scan_tiles<true, false, 4, EMIT, false>;
    return (const void *)ts_scan_tiles<false, false, 4, EMIT, false>;
}
const void *scan_variant(const TsScanParams *p) { return p->emit ? scan_variant_e<true>(p) : scan_variant_e<false>(p); }
}  // namespace

// 1 when the emitting build knows kp.emit == 2 (see TS_READ_INDEX_BUILD)
int ts_k_read_index_built(void) { return TS_READ_INDEX_BUILD; }

int ts_k_prepare(uint32_t lds_bytes) {
    const void *fns[] = {(const void *)ts_scan_tiles<true, true, 4, false, false>, (const void *)ts_scan_tiles<true, false, 4, false, false>,
                         (const void *)ts_scan_tiles<false, false, 4, false, false>, (const void *)ts_scan_tiles<true, true, 6, false, false>,
                         (const void *)ts_scan_tiles<true, true, 4, true, false>, (const void *)ts_scan_tiles<true, false, 4, true, false>,
                         (const void *)ts_scan_tiles<false, false, 4, true, false>, (const void *)ts_scan_tiles<true, true, 6, true, false>,
                         (const void *)ts_scan_tiles<true, true, 4, false, true>, (const void *)ts_scan_tiles<true, true, 6, false, true>,
                         (const void *)ts_scan_tiles<true, true, 4, true, true>, (const void *)ts_scan_tiles<true, true, 6, true, true>};
    // the limit is a property of the function, not of a launch: batches of different geometries share it, so it is
    // raised to the CU's whole LDS rather than set to the size one batch asked for
    const int limit = (int)(lds_bytes > 160u * 1024u ? lds_bytes : 160u * 1024u);
    for (const void *fn : fns) {
        int e = (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, limit);
        if (e) return e;
    }
    return 0;
}

int ts_k_launch_scan(const TsScanParams *p, uint32_t grid, uint32_t lds_bytes, void *stream) {
    const dim3 block(p->waves_per_wg * 64u);
    const void *fn = scan_variant(p);
    if (!fn) return (int)hipErrorInvalidConfiguration;
    void *args[] = {(void *)p};
    return (int)hipLaunchKernel(fn, dim3(grid), block, args, lds_bytes, (hipStream_t)stream);
}

int ts_k_launch_summary(const uint32_t *tile_stats, const uint32_t *seg_first_tile, const uint64_t *seg_nwin,
                        uint32_t nseg, unsigned long long *out, void *stream) {
    if (nseg == 0) return 0;
    hipLaunchKernelGGL(ts_segment_summary, dim3(nseg), dim3(256), 0, (hipStream_t)stream,
                       tile_stats, seg_first_tile, seg_nwin, nseg, out);
    return (int)hipGetLastError();
}

int ts_k_launch_compact(const uint32_t *regions, const uint32_t *wave_fill, const unsigned long long *wave_dense_base,
                        uint32_t region_cap, uint32_t nwaves, uint32_t *dense, int rec16, void *stream) {
    if (nwaves == 0) return 0;
    hipLaunchKernelGGL(ts_compact_regions, dim3(nwaves), dim3(256), 0, (hipStream_t)stream,
                       regions, wave_fill, wave_dense_base, region_cap, nwaves, dense, rec16);
    return (int)hipGetLastError();
}
