// sort.hip - stable LSD radix sort of 64-bit keys (order-preserving transform of the
// float64 residual vector m) with a 32-bit index payload: replaces np.argsort + np.sort
// of src/optim/algorithms.py:92-93 and torch.sort of src/optim/objective.py:74.
// 8 passes of 8 bits.  Every 4096-key tile is ranked with wave ballots (match-any on the digit),
// staged through LDS in digit order and written out in coalesced runs.  Stable, so equal keys
// keep their index order (ties: SURVEY 3.4-e).
// Where a tile's keys go needs the digit counts of all tiles before it.  Default: three kernels
// per pass - (1) per-block digit histogram, (2) scan of the [digit][block] spine (the block that
// finishes last also scans the 256 digit totals), (3) scatter.
// A one-kernel-per-pass variant ("onesweep", Adinets & Merrill 2022: digit histograms of all passes from one
// read, tiles pick up their predecessors' counts by decoupled look-back in ticket order) was built and
// measured in round 2 and removed again: 65 us per pass + 56 us up front = 574 us against 545 us at 6 M keys
// (profiles/r02_sort_onesweep_C2sq_kernel_stats.csv) - the tile kernel loses more to its status traffic
// (2 x 256 agent-scope 8-byte stores per tile across 8 L2s that are not coherent with each other) than the
// histogram pass costs, and batching the look-back 8 deep changed nothing.
#include "rbl_internal.h"
#include <cstdlib>

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 16;                       // keys per thread per tile
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;     // 4096 keys
constexpr int RS_MAX_BLOCKS = 1024;
constexpr int RS_BINS = 256;

struct RsPlan {
    long long n;
    int tiles_per_block;
    int nblocks;
};

RsPlan rs_plan(long long n) {
    RsPlan p;
    p.n = n;
    long long tiles = (n + RS_TILE - 1) / RS_TILE;
    long long tpb = (tiles + RS_MAX_BLOCKS - 1) / RS_MAX_BLOCKS;
    if (tpb < 1) tpb = 1;
    p.tiles_per_block = (int)tpb;
    p.nblocks = (int)((tiles + tpb - 1) / tpb);
    if (p.nblocks < 1) p.nblocks = 1;
    return p;
}

template <typename K>
__device__ inline u32 digit_of(K key, int shift) { return (u32)(key >> shift) & 0xffu; }

// (1) spine[digit * G + block] = number of keys of this block's range with that digit
template <typename K>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const K* __restrict__ keys, long long n, int shift,
                                                         int tiles_per_block, u32* __restrict__ spine, int G) {
    __shared__ u32 h[RS_WAVES][RS_BINS];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < RS_WAVES * RS_BINS; i += RS_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const long long begin = (long long)blockIdx.x * tiles_per_block * RS_TILE;
    long long end = begin + (long long)tiles_per_block * RS_TILE;
    if (end > n) end = n;
    for (long long i = begin + tid; i < end; i += RS_THREADS) atomicAdd(&h[wave][digit_of(keys[i], shift)], 1u);
    __syncthreads();
    for (int d = tid; d < RS_BINS; d += RS_THREADS) {
        u32 c = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) c += h[w][d];
        spine[(long long)d * G + blockIdx.x] = c;
    }
}

// (2a) exclusive scan of each digit's row of the spine (one block per digit, G <= 1024)
// The block that finishes last (an agent-scope counter; bin totals stored and read with agent-scope
// atomics, the per-XCD L2s are not coherent with each other) also does (2b), the exclusive scan of the 256
// digit totals -> bin_base: one launch instead of two.
__global__ __launch_bounds__(1024) void k_rs_scan_rows(u32* __restrict__ spine, int G, u32* bin_total,
                                                        u32* __restrict__ bin_base, u32* done_counter) {
    __shared__ u32 wsum[16];
    __shared__ u32 s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32* row = spine + (long long)blockIdx.x * G;
    u32 x = (tid < G) ? row[tid] : 0u;
    u32 incl = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        u32 y = __shfl_up(incl, off, 64);
        if (lane >= off) incl += y;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (tid < G) row[tid] = base + incl - x;
    if (tid == 1023) {
        __hip_atomic_store(bin_total + blockIdx.x, base + incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u32 prev = __hip_atomic_fetch_add(done_counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (prev == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // (2b) last block: exclusive scan of the digit totals
    __syncthreads();
    u32 t = 0;
    if (tid < RS_BINS) t = __hip_atomic_load(bin_total + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u32 in2 = t;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        u32 y = __shfl_up(in2, off, 64);
        if (lane >= off) in2 += y;
    }
    if (lane == 63 && wave < RS_BINS / 64) wsum[wave] = in2;
    __syncthreads();
    if (tid < RS_BINS) {
        u32 b2 = 0;
        for (int w = 0; w < wave; ++w) b2 += wsum[w];
        bin_base[tid] = b2 + in2 - t;
    }
    if (tid == 0) __hip_atomic_store(done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next pass
}

// (3) scatter.  Wave w of the block owns keys [w*1024, (w+1)*1024) of the tile, 16 rounds
// of 64; ranks among equal digits come from 8 ballots per round plus a per-wave running
// count in LDS (no block barrier inside the round loop).
template <typename K, bool HAS_VAL>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const K* __restrict__ kin, const u32* __restrict__ vin,
                                                            K* __restrict__ kout, u32* __restrict__ vout,
                                                            long long n, int shift, int tiles_per_block,
                                                            const u32* __restrict__ spine,
                                                            const u32* __restrict__ bin_base, int G) {
    __shared__ K skey[RS_TILE];
    __shared__ u32 sval[HAS_VAL ? RS_TILE : 1];
    __shared__ u32 wave_run[RS_WAVES][RS_BINS];  // running digit counts of each wave inside the tile
    __shared__ u32 tile_start[RS_BINS];          // exclusive scan of the tile's digit totals
    __shared__ u32 glob_off[RS_BINS];            // next output slot of every digit for this block
    __shared__ u32 wsum[RS_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (tid < RS_BINS) glob_off[tid] = bin_base[tid] + spine[(long long)tid * G + blockIdx.x];

    for (int t = 0; t < tiles_per_block; ++t) {
        const long long tile_base = ((long long)blockIdx.x * tiles_per_block + t) * RS_TILE;
        if (tile_base >= n) break;
        const long long rem = n - tile_base;
        const int tile_valid = (int)(rem < RS_TILE ? rem : RS_TILE);
        for (int i = tid; i < RS_WAVES * RS_BINS; i += RS_THREADS) (&wave_run[0][0])[i] = 0;
        __syncthreads();

        K key[RS_ITEMS];
        u32 val[RS_ITEMS];
        unsigned short rank[RS_ITEMS];
        volatile u32* myrun = wave_run[wave];
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            const int local = wave * (RS_TILE / RS_WAVES) + r * 64 + lane;
            const long long gi = tile_base + local;
            const bool ok = local < tile_valid;
            key[r] = ok ? kin[gi] : (K)~(K)0;  // padding sorts behind every real key of the tile
            if (HAS_VAL) val[r] = ok ? vin[gi] : 0u;
        }
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            const u32 dg = digit_of(key[r], shift);
            u64 peers = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (dg >> b) & 1u;
                const u64 m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const u32 before = (u32)__popcll(peers & lt_mask);
            const u32 base = myrun[dg];
            __builtin_amdgcn_wave_barrier();
            if (before == 0) myrun[dg] = base + (u32)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
            rank[r] = (unsigned short)(base + before);
        }
        __syncthreads();

        // tile digit totals and their exclusive scan
        u32 tot = 0, pre_w[RS_WAVES];
        if (tid < RS_BINS) {
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) {
                pre_w[w] = tot;
                tot += wave_run[w][tid];
            }
        }
        u32 incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            u32 y = __shfl_up(incl, off, 64);
            if (lane >= off) incl += y;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        if (tid < RS_BINS) {
            u32 base = 0;
            for (int w = 0; w < wave; ++w) base += wsum[w];
            const u32 start = base + incl - tot;
            tile_start[tid] = start;
            // turn the per-wave counts into each wave's first slot of this digit
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) wave_run[w][tid] = start + pre_w[w];
        }
        __syncthreads();

        // local scatter into digit order
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            const u32 dg = digit_of(key[r], shift);
            const u32 pos = wave_run[wave][dg] + rank[r];
            skey[pos] = key[r];
            if (HAS_VAL) sval[pos] = val[r];
        }
        __syncthreads();

        // coalesced write-out: consecutive threads hold consecutive slots of a digit run
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const int p = j * RS_THREADS + tid;
            if (p < tile_valid) {
                const K k = skey[p];
                const u32 dg = digit_of(k, shift);
                const long long dst = (long long)glob_off[dg] + (p - (int)tile_start[dg]);
                kout[dst] = k;
                if (HAS_VAL) vout[dst] = sval[p];
            }
        }
        __syncthreads();
        if (tid < RS_BINS) {
            // padding keys (digit 255 only) are not real output
            u32 real = tot;
            if (tid == RS_BINS - 1) real -= (u32)(RS_TILE - tile_valid);
            glob_off[tid] += real;
        }
        __syncthreads();
    }
}


}  // namespace

size_t sort_ghist_bytes() { return sizeof(u32) * 16; }   // the spine scan's done-counter

size_t sort_spine_bytes() { return sizeof(u32) * RS_BINS * RS_MAX_BLOCKS; }

namespace {
template <typename K>
int radix_passes(SortWorkspace& ws, K* k0, K* k1, int64_t n, bool with_vals, int npass, hipStream_t s) {
    if (n >= (1LL << 32)) {
        rbl_set_error("radix sort: n must be < 2^32");
        return RBL_ERR_INVALID;
    }
    if (!ws.ghist) {
        rbl_set_error("radix sort: workspace without its counter block");
        return RBL_ERR_STATE;
    }
    RsPlan p = rs_plan(n);
    K* kk[2] = {k0, k1};
    int cur = 0;
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = pass * 8;
        hipLaunchKernelGGL(k_rs_hist<K>, dim3(p.nblocks), dim3(RS_THREADS), 0, s, (const K*)kk[cur], (long long)n, shift,
                           p.tiles_per_block, ws.spine, p.nblocks);
        hipLaunchKernelGGL(k_rs_scan_rows, dim3(RS_BINS), dim3(1024), 0, s, ws.spine, p.nblocks, ws.bin_total, ws.bin_base, ws.ghist);
        if (with_vals)
            hipLaunchKernelGGL((k_rs_scatter<K, true>), dim3(p.nblocks), dim3(RS_THREADS), 0, s, (const K*)kk[cur],
                               (const u32*)ws.vals[cur], kk[cur ^ 1], ws.vals[cur ^ 1], (long long)n, shift, p.tiles_per_block,
                               (const u32*)ws.spine, (const u32*)ws.bin_base, p.nblocks);
        else
            hipLaunchKernelGGL((k_rs_scatter<K, false>), dim3(p.nblocks), dim3(RS_THREADS), 0, s, (const K*)kk[cur],
                               (const u32*)nullptr, kk[cur ^ 1], (u32*)nullptr, (long long)n, shift, p.tiles_per_block,
                               (const u32*)ws.spine, (const u32*)ws.bin_base, p.nblocks);
        cur ^= 1;
    }
    RBL_HIP(hipGetLastError());
    return RBL_OK;  // even number of passes: the result is back in the first buffer pair
}
}  // namespace

int launch_radix_sort(SortWorkspace& ws, int64_t n, bool with_vals, hipStream_t s, int key_bits) {
    if (n <= 1) return RBL_OK;
    // keys known to fit key_bits bits need only that many digits; an even number of passes keeps
    // the result in keys[0] / vals[0]
    int npass = (key_bits + 7) / 8;
    if (npass < 1) npass = 1;
    npass = (npass + 1) & ~1;
    if (npass > 8) npass = 8;
    return radix_passes<u64>(ws, ws.keys[0], ws.keys[1], n, with_vals, npass, s);
}

// 32-bit keys (the first n u32 of keys[0], scratch: the first n u32 of keys[1]) with the payload in vals[0]: 4 passes
// over 8 bytes per element instead of 8 passes over 12 (the z-step's fixed-point image of m, elementwise.hip: k_keys32)
int launch_radix_sort32(SortWorkspace& ws, int64_t n, hipStream_t s) {
    if (n <= 1) return RBL_OK;
    return radix_passes<u32>(ws, reinterpret_cast<u32*>(ws.keys[0]), reinterpret_cast<u32*>(ws.keys[1]), n, true, 4, s);
}
