// t5vq_small.hip -- the single-launch forward for SMALL problems (BASELINE config 2: N = 4096, K = 512, D = 768).
//
// At this size a launch costs as much as the work it carries: the chain [distance + argmin] -> [gather + statistics] ->
// [scalars] ran 35 + 11 + 4 us against 20.5 us of matrix work.  Here ONE kernel does the whole forward of
// vqvae/vqvae.py:103-128,147-152,160-162:
//
//   * a workgroup owns 32 tokens against a slice of the codebook (all of it when the grid is large enough without a
//     split); each of its four waves takes the 32 tokens x its own 32 NT codes on v_mfma_f32_32x32x2_f32, operands in
//     the distance kernel's layout and k order, so every dot product is the same fma chain, bit for bit;
//   * every wave stages ITS OWN operand rows (global -> registers, PD chunks ahead -> its private double-buffered LDS
//     image): LDS operations of one wave execute in order, so the main loop has NO workgroup barrier at all -- the
//     distance kernel's 128 x 64 form pays one per 32-deep chunk with a single wave per SIMD to cover it (58 % of peak
//     at C2).  The price: the 32 x 32 token block of a chunk is fetched by all four waves (L1 hits after the first);
//   * |x|^2 is accumulated from the staged rows in row_sqnorm_wave's lane layout and order (bit-identical to
//     t5vq_row_sqnorms_f32), the epilogue is the distance kernel's: d = fl(fl(xx + ee) - 2 dot), first minimum wins;
//   * the S workgroups that share a token tile (code splits) hand their 32 packed keys to the one that finishes last:
//     sc1 (write-through) stores, s_waitcnt vmcnt(0), one agent-scope ticket add -- the workgroup whose add returns
//     S - 1 reads the S x 32 keys back with sc1 loads (MI355X_MICROARCH.md, inter-workgroup visibility, valid forms,
//     row 1: no fence, no polling, nobody waits);
//   * that workgroup gathers its 32 codewords, writes the straight-through output fl(x + fl(q - x)), q - x, the
//     indices, one loss term per token, and adds its counts (one atomic per (workgroup, code): the tokens are grouped
//     in LDS first) and, in training, its rows to embed_sum;
//   * with `scalars` the launch also ends the call: the workgroup that finishes its gather last (a second ticket)
//     reads every count (agent-scope acquire: they were written by atomics) and every loss term (sc1) and writes
//     loss / perplexity in ema_cluster_kernel's own summation orders (bit-identical to t5vq_ema_cluster_f32 mode 0).
//
// Tickets and (with `scalars`) the counts live in a small STATE block that the caller zeroes once; every launch
// leaves it zeroed again (the last arrivers reset what they used), so no call pays a memset.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <type_traits>

#include "t5vq.h"
#include "t5vq_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef T5VQ_SMALL_ABL   // timing-only ablations of the main loop (diagnostic builds; results are wrong with any bit set
#define T5VQ_SMALL_ABL 0 // but 8): 1 no global loads, 2 no ds_writes, 4 no fragment reads, 8 loads spread over all groups
#endif

namespace {

constexpr long long KEY_NONE = 0x7FFFFFFFFFFFFFFFLL;

__device__ __forceinline__ long long pack_key(float d, int idx) {
    int b = __float_as_int(d);
    int s = b >= 0 ? b : (b ^ 0x7FFFFFFF);
    return (static_cast<long long>(s) << 32) | static_cast<unsigned int>(idx);
}
__device__ __forceinline__ int key_index(long long key) { return static_cast<int>(key & 0xFFFFFFFFLL); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float sq4(const f32x4 v, float s) {
    s = fmaf(v.x, v.x, s);
    s = fmaf(v.y, v.y, s);
    s = fmaf(v.z, v.z, s);
    s = fmaf(v.w, v.w, s);
    return s;
}

// every word another workgroup of the launch reads: global address space, agent scope, never flat, never plain
typedef unsigned long long __attribute__((address_space(1))) gu64;
typedef unsigned int __attribute__((address_space(1))) gu32;
#define T5VQ_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

#define T5VQ_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_MFMA = 0x8, SG_VMEM_RD = 0x20, SG_DS_RD = 0x100, SG_DS_WR = 0x200;

constexpr int SM_TOK = 32;   // tokens per workgroup
constexpr int SM_BKC = 32;   // floats of D per stage
constexpr int SM_ROW = 36;   // LDS row stride in floats: 144 B = 9 x 16 B (conflict-free ds_read_b128, as in the scan)
constexpr int SM_MAX_TILES = T5VQ_SMALL_MAX_TOKENS / SM_TOK;

struct SmallArgs {
    const float* x;
    const float* w;
    const float* xx;   // nullable: computed here
    const float* ee;   // |e|^2, +inf padded to ee_len
    int N, K, D, ee_len;
    int n_splits, tiles_per_split;
    long long* slots;        // [n_splits][N] packed keys handed between the code splits of a token tile
    unsigned* tile_ticket;   // [token tiles], zero on entry, zero on exit
    unsigned* done_ticket;   // [1], likewise
    float* out_ste;
    float* out_q;
    float* out_diff;
    long long* idx_out;
    float* counts;       // nullable; zero on entry (the caller's buffer, or the state block's when scalars are fused)
    int reset_counts;    // with scalars: the last workgroup leaves `counts` zeroed again (the state block's invariant)
    float* embed_sum;    // nullable; zero on entry
    float* loss_terms;   // [N], nullable
    float* scalars;      // nullable: [loss, perplexity, 0, raw latent loss] written by the last workgroup
    float* counts_out;   // nullable (only with scalars): the counts copied out before the state block is re-zeroed
    float n_tokens, loss_denom, cc;
    int plain_variant;
    long long* stamps;   // diagnostic builds (-DT5VQ_SMALL_STAMPS): [workgroup][8] wall-clock stamps (100 MHz)
};
#ifdef T5VQ_SMALL_STAMPS
#define T5VQ_STAMP(i)                                                                      \
    do {                                                                                   \
        if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
long long* g_stamps = nullptr;
#else
#define T5VQ_STAMP(i)
#endif

template <int NT, int PD>
struct SmallCfg {
    static_assert(PD == 2 || PD == 4, "register sets / prefetch distance");
    static constexpr int WROWS = 32 * NT;                       // codes per wave per code tile
    static constexpr int CODE_TILE = 4 * WROWS;                 // codes per workgroup per code tile
    static constexpr int STAGE = (SM_TOK + WROWS) * SM_ROW;     // floats per wave per stage
    static constexpr int WAVE_LDS = 2 * STAGE;                  // double buffer
    static constexpr int SHARED_OFF = 4 * WAVE_LDS;             // floats: start of the workgroup-shared tail
    // tail: xx [32] | keys [4][32] int64 | codes [32] int | group masks [32] u64 | group codes [32] int | misc [8] int
    static constexpr int LDS_BYTES = SHARED_OFF * 4 + 32 * 4 + 128 * 8 + 32 * 4 + 32 * 8 + 32 * 4 + 8 * 4 + 16 * 8;
};

// Sum of n loss terms in sum_terms_1024's order (t5vq_kernels.hip), by 256 threads that each play 4 of its 1024:
// virtual thread v = threadIdx.x + 256 m has its real thread's lane and sits in virtual wave (threadIdx.x >> 6) + 4 m,
// so a virtual wave's butterfly is a real wave's.  Loads are sc1 (the terms were handed over inside the launch).
__device__ __forceinline__ double sum_terms_as_1024(const float* p, long long n, double* s /*[16]*/) {
    const gu32* g = (const gu32*)p;
    const long long n4 = n >> 2, tail = n4 << 2;
    double acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const long long v = threadIdx.x + 256 * m;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (long long q = v; q < n4; q += 1024) {
            a0 += static_cast<double>(__uint_as_float(__hip_atomic_load(g + 4 * q, T5VQ_RLX_AGENT)));
            a1 += static_cast<double>(__uint_as_float(__hip_atomic_load(g + 4 * q + 1, T5VQ_RLX_AGENT)));
            a2 += static_cast<double>(__uint_as_float(__hip_atomic_load(g + 4 * q + 2, T5VQ_RLX_AGENT)));
            a3 += static_cast<double>(__uint_as_float(__hip_atomic_load(g + 4 * q + 3, T5VQ_RLX_AGENT)));
        }
        if (tail + v < n) a0 += static_cast<double>(__uint_as_float(__hip_atomic_load(g + tail + v, T5VQ_RLX_AGENT)));
        acc[m] = wave_sum_d((a0 + a1) + (a2 + a3));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) s[(threadIdx.x >> 6) + 4 * m] = acc[m];
    }
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int k = 0; k < 16; ++k) t += s[k];
    return t;
}

// sum_k p_k log(p_k + 1e-10), p = counts / n_tokens, in ema_cluster_kernel<VEC>'s order (K % 4 == 0): thread t of 1024
// owns the float4 groups t, t + 1024, ...; wave butterflies; the 16 wave totals in wave order.  Every thread returns it.
__device__ __forceinline__ double entropy_as_1024(const float* counts, int K, float n_tokens, double* s /*[16]*/) {
    const gu32* g = (const gu32*)counts;
    double acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        double ent = 0.0;
        for (int k4 = threadIdx.x + 256 * m; k4 < (K >> 2); k4 += 1024) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = __uint_as_float(__hip_atomic_load(g + 4 * k4 + e, T5VQ_RLX_AGENT));
                const float p = __fdiv_rn(c, n_tokens);
                ent += static_cast<double>(__fmul_rn(p, logf(__fadd_rn(p, 1e-10f))));
            }
        }
        acc[m] = wave_sum_d(ent);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) s[(threadIdx.x >> 6) + 4 * m] = acc[m];
    }
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += s[i];
    return t;
}

template <int NT, int PD, int NCH>
__global__ __launch_bounds__(256, 1) void small_fwd_kernel(const SmallArgs a) {
    using C = SmallCfg<NT, PD>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WROWS = C::WROWS, CODE_TILE = C::CODE_TILE, STAGE = C::STAGE, ROW = SM_ROW;
    constexpr int NWV = 4 * NT;   // float4 of W per lane per stage (4 of X)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const int r8 = lane >> 3;   // staging: row r8 + 8 j, 16-byte piece c8
    const int c8 = lane & 7;

    float* wl = lds + wave * C::WAVE_LDS;   // this wave's two stages
    float* s_xx = lds + C::SHARED_OFF;                                      // [32]
    long long* s_keys = reinterpret_cast<long long*>(s_xx + 32);            // [4][32]
    int* s_code = reinterpret_cast<int*>(s_keys + 128);                     // [32]
    unsigned long long* s_gmask = reinterpret_cast<unsigned long long*>(s_code + 32);   // [32]
    int* s_gcode = reinterpret_cast<int*>(s_gmask + 32);                    // [32]
    int* s_misc = s_gcode + 32;                                             // [8]: 0 last arriver, 1 groups, 2 last of all
    double* s_red = reinterpret_cast<double*>(s_misc + 8);                  // [16]

    T5VQ_STAMP(0);
    const int N = a.N, K = a.K, D = a.D, S = a.n_splits;
    // blocks b and b + 8 share an XCD (round-robin dispatch; speed only): the S workgroups of one token tile are given
    // ids 8 apart, so their token rows come out of ONE L2
    const int grp = blockIdx.x / (8 * S);
    const int split = (blockIdx.x / 8) % S;
    const int tile = grp * 8 + (blockIdx.x & 7);
    const int tok_base = tile * SM_TOK;
    if (tok_base >= N) return;   // all S workgroups of a tile beyond the batch leave here: its ticket is never touched
    const int n_code_tiles = (K + CODE_TILE - 1) / CODE_TILE;
    const int tile_begin = split * a.tiles_per_split;
    int tile_end = tile_begin + a.tiles_per_split;
    if (tile_end > n_code_tiles) tile_end = n_code_tiles;
    const int n_tiles = tile_end - tile_begin;   // >= 1: the host plans no empty split
    const int n_chunks = D / SM_BKC;             // a multiple of 8 (D % 256 == 0), >= PD

    typedef const f32x4 __attribute__((address_space(1))) * gvec_t;
    typedef const float __attribute__((address_space(1))) * gptr_t;
    const gptr_t xg = (gptr_t)a.x;
    const gptr_t wg = (gptr_t)a.w;
    // staged X rows are taken in an order rotated by the wave id, so that slot 0 is row r8 + 8 wave: the rows whose
    // |x|^2 THIS wave accumulates (a literal register index whatever the wave)
    gptr_t xp[4];
    int x_st[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = r8 + 8 * ((j + wave) & 3);
        int t = tok_base + row;
        if (t > N - 1) t = N - 1;
        xp[j] = xg + static_cast<size_t>(t) * D + 4 * c8;
        x_st[j] = row * ROW + 4 * c8;
    }
    gptr_t wpc[NWV], wpn[NWV];
    auto set_wptrs = [&](gptr_t (&wp)[NWV], int ct) {
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            int c = (tile_begin + ct) * CODE_TILE + wave * WROWS + r8 + 8 * j;
            if (c > K - 1) c = K - 1;
            wp[j] = wg + static_cast<size_t>(c) * D + 4 * c8;
        }
    };
    const int w_st = SM_TOK * ROW + r8 * ROW + 4 * c8;   // + 8 ROW j

    const int tok = tok_base + l31;
    float xv = a.xx ? a.xx[tok < N ? tok : N - 1] : 0.f;
    float xacc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) xacc[m] = 0.f;
    // ring of |x|^2 accumulators: place 0 takes the chunk being staged, then the ring turns by one place, so that chunk
    // kc always meets accumulator kc % 8 (row_sqnorm_wave's lane 8 (kc % 8) + c8) and 8 | n_chunks brings it home
    auto acc_turn = [&](const f32x4 v) {
        const float t0 = sq4(v, xacc[0]);
#pragma unroll
        for (int m = 0; m < 7; ++m) xacc[m] = xacc[m + 1];
        xacc[7] = t0;
    };

    f32x4 sx[PD][4];
    f32x4 sw[PD][NWV];
    set_wptrs(wpc, 0);
    // prologue: chunks 0 .. PD-1 of tile 0 on their way, chunk 0 into stage 0
#pragma unroll
    for (int p = 0; p < PD; ++p) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sx[p][j] = *(gvec_t)(xp[j] + p * SM_BKC);
#pragma unroll
        for (int j = 0; j < NWV; ++j) sw[p][j] = *(gvec_t)(wpc[j] + p * SM_BKC);
    }
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(wl + x_st[j]) = sx[0][j];
#pragma unroll
        for (int j = 0; j < NWV; ++j) *reinterpret_cast<f32x4*>(wl + w_st + 8 * ROW * j) = sw[0][j];
        acc_turn(sx[0][0]);
    }

    const int rd_x = l31 * ROW + 4 * h;
    const int rd_w = SM_TOK * ROW + l31 * ROW + 4 * h;
    struct Frag {
        f32x4 b;
        f32x4 a[NT];
    };
    Frag f[2];
    auto read_frag = [&](Frag& fr, const float* base, int g) {
        fr.b = *reinterpret_cast<const f32x4*>(base + rd_x + 8 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t) fr.a[t] = *reinterpret_cast<const f32x4*>(base + rd_w + t * 32 * ROW + 8 * g);
    };
    read_frag(f[0], wl, 0);   // same wave wrote it: LDS operations of one wave execute in order
#ifdef T5VQ_SMALL_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    T5VQ_STAMP(1);
#endif

    int buf = 0;
    f32x16 acc[NT];
    float best = __int_as_float(0x7F800000);
    int best_code = (tile_begin * CODE_TILE + wave * WROWS);

    using NoAcc = std::integral_constant<int, 0>;
    using DoAcc = std::integral_constant<int, 1>;
    // One chunk: MFMAs on stage `buf`; global loads of the chunk PD ahead (pointers wp, column koff) into register set P;
    // the next chunk (set P + 1) written to the other stage; first fragments of the next chunk read.
    auto body = [&](const gptr_t (&wp)[NWV], int koff, auto parity, auto accm) {
        constexpr int P = decltype(parity)::value;
        constexpr int PN = (P + 1) % PD;
        constexpr bool AM = decltype(accm)::value != 0;
        const float* cur = wl + buf * STAGE;
        float* nxt = wl + (buf ^ 1) * STAGE;
        // set P's old contents (this chunk) went to LDS one chunk ago
#if !(T5VQ_SMALL_ABL & 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) sx[P][j] = *(gvec_t)(xp[j] + koff);
#pragma unroll
        for (int j = 0; j < NWV; ++j) sw[P][j] = *(gvec_t)(wp[j] + koff);
#endif
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#if !(T5VQ_SMALL_ABL & 4)
            if (g < 3) read_frag(f[(g + 1) & 1], cur, g + 1);
#else
            if (g < 3) f[(g + 1) & 1] = f[g & 1];
#endif
            if (g == 2) {
#if !(T5VQ_SMALL_ABL & 2)
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(nxt + x_st[j]) = sx[PN][j];
#pragma unroll
                for (int j = 0; j < NWV; ++j) *reinterpret_cast<f32x4*>(nxt + w_st + 8 * ROW * j) = sw[PN][j];
#else
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(sx[PN][j]));
#pragma unroll
                for (int j = 0; j < NWV; ++j) asm volatile("" ::"v"(sw[PN][j]));
#endif
                if constexpr (AM) acc_turn(sx[PN][0]);
            }
#if !(T5VQ_SMALL_ABL & 4)
            if (g == 3) read_frag(f[0], nxt, 0);
#else
            if (g == 3) f[0] = f[1];
#endif
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[g & 1].a[t][j], f[g & 1].b[j], acc[t], 0, 0, 0);
        }
        // ---- issue order: per k-group 4 NT MFMAs; loads behind the MFMAs of groups 0-1, ds_writes behind group 2's,
        //      fragment reads one per MFMA at the head of each group
        constexpr int M = 4 * NT, R = NT + 1, NLD = 4 + NWV;
#if T5VQ_SMALL_ABL == 0 || T5VQ_SMALL_ABL == 8
#if T5VQ_SMALL_ABL == 8
        // loads spread evenly over the four groups
        constexpr int LG = (NLD + 3) / 4, Q = (NLD + M - 1) / M;
        static_assert(LG <= M, "one load per MFMA at most");
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            T5VQ_SGB(SG_DS_RD, R);
#pragma unroll
            for (int i = 0; i < M; ++i) {
                T5VQ_SGB(SG_MFMA, 1);
                if (i % 2 == 1 && i / 2 < (NLD - LG * g < LG ? NLD - LG * g : LG)) T5VQ_SGB(SG_VMEM_RD, 1);
                if (g == 2) T5VQ_SGB(SG_DS_WR, Q);
            }
        }
#else
        constexpr int L0 = (NLD + 1) / 2, L1 = NLD - L0, Q = (NLD + M - 1) / M;
        static_assert(L0 <= M && L1 <= M, "one load per MFMA at most");
        T5VQ_SGB(SG_DS_RD, R);            // group 0: fragments of group 1, half of the loads
#pragma unroll
        for (int i = 0; i < M; ++i) {
            T5VQ_SGB(SG_MFMA, 1);
            if (i < L0) T5VQ_SGB(SG_VMEM_RD, 1);
        }
        T5VQ_SGB(SG_DS_RD, R);            // group 1: fragments of group 2, the other half
#pragma unroll
        for (int i = 0; i < M; ++i) {
            T5VQ_SGB(SG_MFMA, 1);
            if (i < L1) T5VQ_SGB(SG_VMEM_RD, 1);
        }
        T5VQ_SGB(SG_DS_RD, R);            // group 2: fragments of group 3, the ds_writes spread over its MFMAs
#pragma unroll
        for (int i = 0; i < M; ++i) {
            T5VQ_SGB(SG_MFMA, 1);
            T5VQ_SGB(SG_DS_WR, Q);
        }
        T5VQ_SGB(SG_DS_RD, R);            // group 3: first fragments of the next chunk
        T5VQ_SGB(SG_MFMA, M);
#endif
#endif
        buf ^= 1;
    };

    // |e|^2 of this lane's codes in the tile, fetched at the top of the tile (the epilogue used to wait for them)
    auto load_ee = [&](f32x4 (&e4)[NT][4], int ct) {
        const int code0 = (tile_begin + ct) * CODE_TILE + wave * WROWS + 4 * h;
        const f32x4 inf4 = {__int_as_float(0x7F800000), __int_as_float(0x7F800000), __int_as_float(0x7F800000),
                            __int_as_float(0x7F800000)};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int cb = code0 + t * 32 + 8 * g4;
                e4[t][g4] = cb + 3 < a.ee_len ? *reinterpret_cast<const f32x4*>(a.ee + cb) : inf4;
            }
    };

    for (int ct = 0; ct < n_tiles; ++ct) {
        set_wptrs(wpn, ct + 1 < n_tiles ? ct + 1 : ct);
        f32x4 e4[NT][4];
        load_ee(e4, ct);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // chunks 0 .. n_chunks - PD - 1 stage a chunk of this tile; the last PD ones stage chunks 0 .. PD - 1 of the next
        // tile (a harmless reload on the last tile); the tile's very last body writes chunk 0 of the NEXT pass over the
        // token rows to LDS, which must not reach the |x|^2 accumulators
        for (int kc = 0; kc + PD < n_chunks; kc += PD) {
            if constexpr (PD == 2) {
                body(wpc, (kc + 2) * SM_BKC, std::integral_constant<int, 0>(), DoAcc());
                body(wpc, (kc + 3) * SM_BKC, std::integral_constant<int, 1>(), DoAcc());
            } else {
                body(wpc, (kc + 4) * SM_BKC, std::integral_constant<int, 0>(), DoAcc());
                body(wpc, (kc + 5) * SM_BKC, std::integral_constant<int, 1>(), DoAcc());
                body(wpc, (kc + 6) * SM_BKC, std::integral_constant<int, 2>(), DoAcc());
                body(wpc, (kc + 7) * SM_BKC, std::integral_constant<int, 3>(), DoAcc());
            }
        }
        if constexpr (PD == 2) {
            body(wpn, 0, std::integral_constant<int, 0>(), DoAcc());
            body(wpn, SM_BKC, std::integral_constant<int, 1>(), NoAcc());
        } else {
            body(wpn, 0, std::integral_constant<int, 0>(), DoAcc());
            body(wpn, SM_BKC, std::integral_constant<int, 1>(), DoAcc());
            body(wpn, 2 * SM_BKC, std::integral_constant<int, 2>(), DoAcc());
            body(wpn, 3 * SM_BKC, std::integral_constant<int, 3>(), NoAcc());
        }
#pragma unroll
        for (int j = 0; j < NWV; ++j) wpc[j] = wpn[j];
        if (ct == n_tiles - 1) T5VQ_STAMP(2);

        if (ct == 0 && !a.xx) {
            // wave_sum's xor butterfly: lane bits 5..3 = the 8 accumulators (ring at home), bits 2..0 = the 8 lanes of a row
            const float n0 = xacc[0] + xacc[4], n1 = xacc[1] + xacc[5];
            const float n2 = xacc[2] + xacc[6], n3 = xacc[3] + xacc[7];
            const float p0 = n0 + n2, p1 = n1 + n3;
            float q = p0 + p1;
            q += __shfl_xor(q, 4);
            q += __shfl_xor(q, 2);
            q += __shfl_xor(q, 1);
            if (c8 == 0) s_xx[r8 + 8 * wave] = q;
            __syncthreads();
            xv = s_xx[l31];
        }

        const int code0 = (tile_begin + ct) * CODE_TILE + wave * WROWS + 4 * h;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int code = code0 + t * 32 + 8 * g4 + c;
                    const float s = xv + e4[t][g4][c];
                    const float d = fmaf(-2.0f, acc[t][g4 * 4 + c], s);
                    if (d < best) {
                        best = d;
                        best_code = code;
                    }
                }
    }

    // lane halves, then the four waves: packed-key minimum = lowest distance, then lowest code
    long long key = pack_key(best, best_code);
    {
        const long long other = __shfl_xor(key, 32);
        key = other < key ? other : key;
    }
    if (h == 0) s_keys[wave * 32 + l31] = key;
    __syncthreads();
    T5VQ_STAMP(3);
    const bool mine = tid < SM_TOK && tok_base + tid < N;
    const int my_tok = tok_base + tid;
    if (tid < SM_TOK) {
        key = s_keys[tid];
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) {
            const long long o = s_keys[wv * 32 + tid];
            key = o < key ? o : key;
        }
    }
    if (S > 1) {
        if (mine)
            __hip_atomic_store((gu64*)(a.slots + static_cast<size_t>(split) * N + my_tok),
                               static_cast<unsigned long long>(key), T5VQ_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the storing wave drains before it signals
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add((gu32*)(a.tile_ticket + tile), 1u, T5VQ_RLX_AGENT);
            const int last = old == static_cast<unsigned>(S - 1);
            if (last) __hip_atomic_store((gu32*)(a.tile_ticket + tile), 0u, T5VQ_RLX_AGENT);   // reset for the next launch
            s_misc[0] = last;
        }
        __syncthreads();
        if (!s_misc[0]) return;
        if (mine) {
            for (int s = 0; s < S; ++s) {
                if (s == split) continue;
                const long long o = static_cast<long long>(
                    __hip_atomic_load((gu64*)(a.slots + static_cast<size_t>(s) * N + my_tok), T5VQ_RLX_AGENT));
                key = o < key ? o : key;
            }
        }
    }
    if (tid < SM_TOK) {
        int code = -1;
        if (mine) {
            code = key_index(key);
            code = code < 0 ? 0 : (code > K - 1 ? K - 1 : code);
            if (a.idx_out) a.idx_out[my_tok] = code;
        }
        s_code[tid] = code;
    }
    __syncthreads();
    T5VQ_STAMP(4);

    // ---- gather + straight-through + loss terms: wave w takes tokens 8 w .. 8 w + 7, all rows requested at once
    {
        const f32x4* xr[8];
        f32x4 xq[8][NCH], qq[8][NCH];
        int codes[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = tok_base + 8 * wave + i;
            codes[i] = s_code[8 * wave + i];
            const int tc = t < N ? t : N - 1;
            const int cc_ = codes[i] < 0 ? 0 : codes[i];
            xr[i] = reinterpret_cast<const f32x4*>(a.x + static_cast<size_t>(tc) * D);
            const f32x4* wr = reinterpret_cast<const f32x4*>(a.w + static_cast<size_t>(cc_) * D);
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                xq[i][j] = xr[i][j * 64 + lane];
                qq[i][j] = wr[j * 64 + lane];
            }
        }
        float* trow = wl;   // the wave's staging area is free now: transposition row for the embed_sum flush
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = tok_base + 8 * wave + i;
            if (t >= N) continue;   // wave-uniform
            const size_t rowoff = static_cast<size_t>(t) * D;
            float lacc = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c4 = j * 64 + lane;
                const f32x4 df = qq[i][j] - xq[i][j];
                if (a.out_ste) reinterpret_cast<f32x4*>(a.out_ste + rowoff)[c4] = xq[i][j] + df;
                if (a.out_q) reinterpret_cast<f32x4*>(a.out_q + rowoff)[c4] = qq[i][j];
                if (a.out_diff) reinterpret_cast<f32x4*>(a.out_diff + rowoff)[c4] = df;
                lacc = fmaf(df.x, df.x, lacc);
                lacc = fmaf(df.y, df.y, lacc);
                lacc = fmaf(df.z, df.z, lacc);
                lacc = fmaf(df.w, df.w, lacc);
            }
            if (a.loss_terms) {
                lacc = wave_sum(lacc);
                if (lane == 0)
                    __hip_atomic_store((gu32*)(a.loss_terms + t), __float_as_uint(lacc), T5VQ_RLX_AGENT);
            }
            if (a.embed_sum) {
#pragma unroll
                for (int j = 0; j < NCH; ++j) reinterpret_cast<f32x4*>(trow)[j * 64 + lane] = xq[i][j];
                float* dst = a.embed_sum + static_cast<size_t>(codes[i]) * D;
#pragma unroll
                for (int mcol = 0; mcol < NCH * 4; ++mcol) atomicAdd(dst + mcol * 64 + lane, trow[mcol * 64 + lane]);
            }
        }
    }
    // counts: one atomic per (workgroup, code) -- the 32 tokens are grouped in LDS by wave 0
    if (a.counts && wave == 0) {
        const int mine_c = lane < SM_TOK ? s_code[lane] : -1;
        unsigned long long m = 0;
        for (int t = 0; t < SM_TOK; ++t)
            if (s_code[t] == mine_c) m |= 1ull << t;
        if (mine_c >= 0 && (__ffsll(static_cast<long long>(m)) - 1) == lane)
            atomicAdd(a.counts + mine_c, static_cast<float>(__popcll(m)));
    }
    T5VQ_STAMP(5);
    if (!a.scalars) return;

    // ---- the call's scalars, by the workgroup whose gather finishes last
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its loss terms and count atomics have left
    __syncthreads();
    if (tid == 0) {
        const unsigned n_tok_tiles = static_cast<unsigned>((N + SM_TOK - 1) / SM_TOK);
        const unsigned old = __hip_atomic_fetch_add((gu32*)a.done_ticket, 1u, T5VQ_RLX_AGENT);
        const int last = old == n_tok_tiles - 1;
        if (last) {
            __hip_atomic_store((gu32*)a.done_ticket, 0u, T5VQ_RLX_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the counts were written by atomics
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s_misc[2] = last;
    }
    __syncthreads();
    if (!s_misc[2]) return;
    T5VQ_STAMP(6);
    const double ent = entropy_as_1024(a.counts, K, a.n_tokens, s_red);
    const double lsum = a.loss_terms ? sum_terms_as_1024(a.loss_terms, N, s_red) : 0.0;
    // every thread has read its counts (the barriers inside the two sums): copy them out if asked, and leave the state
    // block's counts zeroed for the next call
    for (int k = tid; k < K; k += 256) {
        if (a.counts_out) a.counts_out[k] = __uint_as_float(__hip_atomic_load((const gu32*)(a.counts + k), T5VQ_RLX_AGENT));
        if (a.reset_counts) a.counts[k] = 0.f;
    }
    if (tid == 0) {
        const float m = a.loss_terms ? __fdiv_rn(static_cast<float>(lsum), a.loss_denom) : 0.f;
        const float cm = __fmul_rn(a.cc, m);
        a.scalars[0] = a.plain_variant ? __fadd_rn(m, cm) : cm;
        a.scalars[1] = expf(static_cast<float>(-ent));
        a.scalars[2] = 0.f;
        a.scalars[3] = m;
    }
    T5VQ_STAMP(7);
}


inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

constexpr int64_t STATE_TICKET_BYTES = 4 * SM_MAX_TILES + 64;   // tile tickets, then the done ticket on a line of its own
constexpr int64_t SMALL_KEY_SLOTS = 65536;                       // splits x N never exceeds this (the workspace's key region)

struct SmallPlan {
    int nt, splits, tiles_per_split;
    int64_t blocks;
};

// Tile and split: the 32 x 256 tile (NT = 2) when it alone yields a workgroup per CU, else 32 x 128; the codebook is cut
// into the most splits that keep the grid within one workgroup per CU (a split costs a hand-off, a second round of
// workgroups would cost a whole tile time).
bool plan_small(int64_t N, int64_t K, SmallPlan* p) {
    const int64_t tok_tiles = (N + SM_TOK - 1) / SM_TOK;
    const int64_t ct256 = (K + 255) / 256, ct128 = (K + 127) / 128;
    p->nt = tok_tiles * ct256 >= 256 ? 2 : 1;
    const int64_t code_tiles = p->nt == 2 ? ct256 : ct128;
    int64_t want = 256 / tok_tiles;
    if (want < 1) want = 1;
    int64_t s = code_tiles < want ? code_tiles : want;
    const int64_t tps = (code_tiles + s - 1) / s;
    s = (code_tiles + tps - 1) / tps;   // no empty split
    p->splits = static_cast<int>(s);
    p->tiles_per_split = static_cast<int>(tps);
    p->blocks = (tok_tiles + 7) / 8 * 8 * s;
    return s * N <= SMALL_KEY_SLOTS;
}

bool small_shape(int64_t N, int64_t K, int64_t D) {
    SmallPlan p;
    return N > 0 && K > 0 && N < T5VQ_SMALL_MAX_TOKENS && D % 256 == 0 && D <= 1024 &&
           ((N + 127) / 128) * ((K + 127) / 128) < 256 && plan_small(N, K, &p);
}

int g_small_pd() {   // prefetch distance in chunks (register sets): T5VQ_SMALL_PD = 2 | 4, for A/B timing
    static const int pd = [] {
        const char* e = getenv("T5VQ_SMALL_PD");
        return (e && e[0] == '4') ? 4 : 2;
    }();
    return pd;
}

template <int NT, int PD, int NCH>
int launch_small_t(const SmallArgs& a, const SmallPlan& plan, hipStream_t stream) {
    using C = SmallCfg<NT, PD>;
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    T5VQ_I_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_done[dev].load(std::memory_order_acquire)) {
        T5VQ_I_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&small_fwd_kernel<NT, PD, NCH>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((small_fwd_kernel<NT, PD, NCH>), dim3(static_cast<unsigned>(plan.blocks)), dim3(256), C::LDS_BYTES,
                       stream, a);
    T5VQ_I_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

template <int NT, int PD>
int launch_small_nch(const SmallArgs& a, const SmallPlan& plan, hipStream_t stream) {
    switch (a.D / 256) {
        case 1: return launch_small_t<NT, PD, 1>(a, plan, stream);
        case 2: return launch_small_t<NT, PD, 2>(a, plan, stream);
        case 3: return launch_small_t<NT, PD, 3>(a, plan, stream);
        default: return launch_small_t<NT, PD, 4>(a, plan, stream);
    }
}

}  // namespace

extern "C" {

int64_t t5vq_small_state_bytes(int64_t K) {
    if (K < 0) return 0;
    return align_up(STATE_TICKET_BYTES + 4 * align_up(K, 4), 256);
}

int t5vq_vq_forward_small_supported(int64_t N, int64_t K, int64_t D) { return small_shape(N, K, D) ? 1 : 0; }

int t5vq_vq_forward_small_f32(const float* x, const float* w, const float* xx_in, const float* ee_in, int64_t N,
                              int64_t K, int64_t D, void* workspace, int64_t workspace_bytes, void* state,
                              float* out_ste, float* out_q, float* out_diff, int64_t* idx_out, float* counts,
                              float* embed_sum, float* loss_terms, float* scalars, float loss_denom,
                              float commitment_cost, int plain_variant, t5vq_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    T5VQ_I_REQUIRE(small_shape(N, K, D), "shape outside the single-launch small-shape forward (t5vq_vq_forward_small_supported)");
    T5VQ_I_REQUIRE(x && w && workspace && state, "null pointer");
    T5VQ_I_REQUIRE(aligned16(x) && aligned16(w) && aligned16(workspace) && aligned16(state) && (!ee_in || aligned16(ee_in)),
                   "x, w, workspace, state, ee_in must be 16-byte aligned");
    T5VQ_I_REQUIRE((!out_ste || aligned16(out_ste)) && (!out_q || aligned16(out_q)) && (!out_diff || aligned16(out_diff)),
                   "outputs must be 16-byte aligned");
    T5VQ_I_REQUIRE(workspace_bytes >= t5vq_forward_workspace_bytes(N, K), "workspace smaller than t5vq_forward_workspace_bytes(N, K)");
    T5VQ_I_REQUIRE(!scalars || K % 4 == 0, "scalars in the same launch need K % 4 == 0");
    T5VQ_I_REQUIRE(!scalars || !embed_sum, "scalars in the same launch are for calls without an EMA update");
    SmallPlan plan;
    plan_small(N, K, &plan);
    // workspace layout = t5vq_vq_forward_f32's (FwdWorkspace in t5vq_kernels.hip): key slots first, then |x|^2 (unused
    // here), |e|^2, loss terms
    char* base = static_cast<char*>(workspace);
    const int64_t off_xx = align_up(8 * SMALL_KEY_SLOTS, 16);
    const int64_t off_ee = align_up(off_xx + 4 * N, 16);
    const int64_t off_terms = align_up(off_ee + 4 * t5vq_padded_codes(K), 16);
    const float* ee = ee_in;
    if (!ee) {
        float* e = reinterpret_cast<float*>(base + off_ee);
        const int rc = t5vq_row_sqnorms_f32(nullptr, 0, nullptr, w, K, e, D, stream_);
        if (rc != T5VQ_OK) return rc;
        ee = e;
    }
    char* st = static_cast<char*>(state);
    SmallArgs a{};
    a.x = x, a.w = w, a.xx = xx_in, a.ee = ee;
    a.N = static_cast<int>(N), a.K = static_cast<int>(K), a.D = static_cast<int>(D);
    a.ee_len = static_cast<int>(t5vq_padded_codes(K));
    a.n_splits = plan.splits, a.tiles_per_split = plan.tiles_per_split;
    a.slots = reinterpret_cast<long long*>(base);
    a.tile_ticket = reinterpret_cast<unsigned*>(st);
    a.done_ticket = reinterpret_cast<unsigned*>(st + 4 * SM_MAX_TILES);
    a.out_ste = out_ste, a.out_q = out_q, a.out_diff = out_diff;
    a.idx_out = reinterpret_cast<long long*>(idx_out);
    a.embed_sum = embed_sum;
    a.loss_terms = loss_terms ? loss_terms : (scalars ? reinterpret_cast<float*>(base + off_terms) : nullptr);
    a.scalars = scalars;
    a.n_tokens = static_cast<float>(N), a.loss_denom = loss_denom, a.cc = commitment_cost;
    a.plain_variant = plain_variant;
    if (scalars) {   // the counts of the call live in the state block (zero between calls) and are copied out if asked
        a.counts = reinterpret_cast<float*>(st + STATE_TICKET_BYTES);
        a.reset_counts = 1;
        a.counts_out = counts;
    } else {
        a.counts = counts;
        if (counts) T5VQ_I_CHECK_HIP(hipMemsetAsync(counts, 0, static_cast<size_t>(K) * 4, stream));
    }
    if (embed_sum) T5VQ_I_CHECK_HIP(hipMemsetAsync(embed_sum, 0, static_cast<size_t>(K) * D * 4, stream));
#ifdef T5VQ_SMALL_STAMPS
    a.stamps = g_stamps;
#endif
    const int pd = g_small_pd();
    if (plan.nt == 2) return pd == 4 ? launch_small_nch<2, 4>(a, plan, stream) : launch_small_nch<2, 2>(a, plan, stream);
    return pd == 4 ? launch_small_nch<1, 4>(a, plan, stream) : launch_small_nch<1, 2>(a, plan, stream);
}

#ifdef T5VQ_SMALL_STAMPS
void t5vq_small_set_stamps(void* p) { g_stamps = static_cast<long long*>(p); }
#endif

}  // extern "C"
