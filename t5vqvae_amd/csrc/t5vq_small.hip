// t5vq_small.hip -- the single-launch forward for SMALL problems (BASELINE config 2: N = 4096, K = 512, D = 768).
//
// At this size a launch costs as much as the work it carries: the chain [distance + argmin] -> [gather + statistics] ->
// [scalars] ran 35 + 11 + 4 us against 20.5 us of matrix work.  Here ONE kernel does the whole forward of
// vqvae/vqvae.py:103-128,147-152,160-162:
//
//   * a workgroup owns 32 tokens against a slice of the codebook (all of it when the grid is large enough without a
//     split); each of its four waves takes the 32 tokens x its own 32 NT codes on v_mfma_f32_32x32x2_f32, operands in
//     the distance kernel's layout and k order, so every dot product is the same fma chain, bit for bit;
//   * every wave stages ITS OWN operand rows by LDS-DMA (global_load_lds_dwordx4, two chunks ahead, a ring of three
//     private stages): a wave reads only what it fetched itself, so its own vmcnt wait is all the ordering the main loop
//     needs -- NO workgroup barrier, no VGPR round trip, no ds_write.  The price: the 32 x 32 token block of a chunk is
//     fetched by all four waves (L1 hits after the first);
//   * |x|^2 is summed from the token rows themselves, row_sqnorm_wave's way (bit-identical to t5vq_row_sqnorms_f32),
//     the epilogue is the distance kernel's: d = fl(fl(xx + ee) - 2 dot), first minimum wins;
//   * the S workgroups that share a token tile (code splits) hand their 32 packed keys over inside the launch: sc1
//     (write-through) stores, s_waitcnt vmcnt(0), one agent-scope ticket add; whoever finishes last reads the keys back
//     with sc1 loads (MI355X_MICROARCH.md, inter-workgroup visibility, valid forms, row 1: no fence).  Two splits that
//     meet within a bounded poll share the gather, 16 tokens each; nobody waits without a bound;
//   * the finishing workgroup(s) gather the codewords, write the straight-through output fl(x + fl(q - x)), q - x, the
//     indices, one loss term per token, and add their counts (one atomic per (workgroup, code): the tokens are grouped
//     in LDS first) and, in training, their rows to embed_sum;
//   * with `scalars` the launch also ends the call: loss terms and codes leave first (sc1) and are signalled on a second
//     ticket before the bulk stores; the workgroup whose signal completes the batch builds the histogram in LDS and
//     writes loss / perplexity in ema_cluster_kernel's own summation orders (bit-identical to t5vq_ema_cluster_f32
//     mode 0) -- no counts in global memory, no acquire.
//
// The tickets live in a small STATE block that the caller zeroes once; every launch leaves it zeroed again (whoever
// touches a ticket last resets it), so no call pays a memset.
//
// Measured at C2 (tools/experiments/small_stamps.py, back to back): prologue 1.9 us, main loop 27.5 us (the same loop
// with no staging at all: 23.0 us = 49152 MFMA cycles at the 2.14 GHz the chip holds here), epilogue + merge 2.2,
// hand-off 1.9, gather 3.7-5, scalars 3-4: 45.0 us per call against 45.7 us for the three-launch chain.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <type_traits>

#include "t5vq.h"
#include "t5vq_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef T5VQ_SMALL_ABL   // timing-only ablations of the main loop (diagnostic builds; results are wrong unless 0):
#define T5VQ_SMALL_ABL 0 // 1 no DMA in the loop, 2 every chunk fetches the same columns, 4 token rows not fetched
#endif

namespace {

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


// s_waitcnt that waits for vmcnt <= n only (expcnt 7, lgkmcnt 15): vmcnt's low four bits sit in [3:0], the high two in [15:14]
constexpr int vmcnt_only(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }

constexpr int SM_TOK = 32;   // tokens per workgroup
constexpr int SM_BKC = 32;   // floats of D per stage
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
    float* counts;       // nullable; without scalars: zero on entry, added to; with scalars: written by the last workgroup
    float* embed_sum;    // nullable; zero on entry
    float* loss_terms;   // [N], nullable
    float* scalars;      // nullable: [loss, perplexity, 0, raw latent loss] written by the last workgroup
    int* codes;          // [N] (only with scalars): every token's code, handed to the workgroup that finishes last
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
    static constexpr int WROWS = 32 * NT;                       // codes per wave per code tile
    static constexpr int CODE_TILE = 4 * WROWS;                 // codes per workgroup per code tile
    static constexpr int STAGE = (SM_TOK + WROWS) * SM_BKC;     // floats per wave per stage: unpadded 128-byte rows
    static constexpr int WAVE_LDS = 3 * STAGE;                  // ring of three stages: multiplied, landed / landing, being filled
    static constexpr int SHARED_OFF = 4 * WAVE_LDS;             // floats: start of the workgroup-shared tail
    // tail: xx [32] | keys [4][32] int64 | codes [32] int | group masks [32] u64 | group codes [32] int | misc [8] int
    static constexpr int LDS_BYTES = SHARED_OFF * 4 + 32 * 4 + 128 * 8 + 32 * 4 + 32 * 8 + 32 * 4 + 8 * 4 + 16 * 8;
};

// 16-byte sc1 loads of words handed over inside the launch (the write-through stores of other workgroups): four in
// flight per thread, one wait.  The wait statement takes the registers as operands, so no use can be scheduled before it.
__device__ __forceinline__ void load4_sc1(const float* p0, const float* p1, const float* p2, const float* p3, bool k0,
                                          bool k1, bool k2, bool k3, f32x4 (&u)[4]) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    u[0] = u[1] = u[2] = u[3] = z;
    if (k0) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(u[0]) : "v"(p0) : "memory");
    if (k1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(u[1]) : "v"(p1) : "memory");
    if (k2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(u[2]) : "v"(p2) : "memory");
    if (k3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(u[3]) : "v"(p3) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]) : : "memory");
}

// Sum of n loss terms in sum_terms_1024's order (t5vq_kernels.hip), by 256 threads that each play 4 of its 1024:
// virtual thread v = threadIdx.x + 256 m has its real thread's lane and sits in virtual wave (threadIdx.x >> 6) + 4 m,
// so a virtual wave's butterfly is a real wave's.  Virtual thread v owns the groups of four terms v, v + 1024, ...;
// accumulator e takes element e of each group (a group that does not exist adds +0: exact).
__device__ __forceinline__ double sum_terms_as_1024(const float* p, long long n, double* s /*[16]*/) {
    const long long n4 = n >> 2, tail = n4 << 2;
    double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0},
           a3[4] = {0.0, 0.0, 0.0, 0.0};
    for (long long q0 = threadIdx.x; q0 < n4; q0 += 1024) {   // uniform trip count up to the last partial round
        f32x4 u[4];
        load4_sc1(p + 4 * q0, p + 4 * (q0 + 256), p + 4 * (q0 + 512), p + 4 * (q0 + 768), true, q0 + 256 < n4,
                  q0 + 512 < n4, q0 + 768 < n4, u);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            a0[m] += static_cast<double>(u[m].x);
            a1[m] += static_cast<double>(u[m].y);
            a2[m] += static_cast<double>(u[m].z);
            a3[m] += static_cast<double>(u[m].w);
        }
    }
    double acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const long long v = threadIdx.x + 256 * m;
        if (tail + v < n)
            a0[m] += static_cast<double>(__uint_as_float(__hip_atomic_load((const gu32*)p + tail + v, T5VQ_RLX_AGENT)));
        acc[m] = wave_sum_d((a0[m] + a1[m]) + (a2[m] + a3[m]));
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
// The counts are the integers of an LDS histogram (exact in fp32, as the atomically added float counts are).
__device__ __forceinline__ double entropy_as_1024(const int* hist, int K, float n_tokens, double* s /*[16]*/) {
    const int k4n = K >> 2;
    double acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        double ent = 0.0;
        for (int k4 = threadIdx.x + 256 * m; k4 < k4n; k4 += 1024) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pr = __fdiv_rn(static_cast<float>(hist[4 * k4 + e]), n_tokens);
                ent += static_cast<double>(__fmul_rn(pr, logf(__fadd_rn(pr, 1e-10f))));
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
    constexpr int WROWS = C::WROWS, CODE_TILE = C::CODE_TILE, STAGE = C::STAGE;
    constexpr int NWV = 4 * NT;   // float4 of W per lane per stage (4 of X)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS stage addresses stay in SGPRs (M0)
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const int r8 = lane >> 3;   // staging: row r8 + 8 j, 16-byte piece c8
    const int c8 = lane & 7;

    float* wl = lds + wave * C::WAVE_LDS;   // this wave's ring of stages
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

    typedef const float __attribute__((address_space(1))) * gptr_t;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const gptr_t xg = (gptr_t)a.x;
    const gptr_t wg = (gptr_t)a.w;
    // LDS-DMA staging (global_load_lds_dwordx4): lane L of a wave-instruction fills the 16 bytes at base + 16 L = row
    // L / 8, slot L % 8 of an 8-row slab of unpadded 128-byte rows; slot s of row r holds piece s ^ ((r >> 1) & 7) of the
    // row's eight 16-byte pieces -- the bank swizzle sits on the SOURCE address.  No VGPR round trip, no ds_write: the
    // register form of this loop spent its issue slots on 12 ds_write_b128 per chunk (13 cycles each on the store path).
    gptr_t xp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int t = tok_base + 8 * j + r8;
        if (t > N - 1) t = N - 1;
        xp[j] = xg + static_cast<size_t>(t) * D + 4 * (c8 ^ (((r8 >> 1) + 4 * j) & 7));
    }
    gptr_t wpc[NWV], wpn[NWV];
    auto set_wptrs = [&](gptr_t (&wp)[NWV], int ct) {
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            int c = (tile_begin + ct) * CODE_TILE + wave * WROWS + r8 + 8 * j;
            if (c > K - 1) c = K - 1;
            wp[j] = wg + static_cast<size_t>(c) * D + 4 * (c8 ^ (((r8 >> 1) + 4 * j) & 7));   // row 32 + 8 j + r8
        }
    };
    const int tok = tok_base + l31;
    float xv = a.xx ? a.xx[tok < N ? tok : N - 1] : 0.f;
    set_wptrs(wpc, 0);
    float* st0 = wl;                 // stage of the chunk being multiplied
    float* st1 = wl + STAGE;         // the next chunk's
    float* st2 = wl + 2 * STAGE;     // the one being filled

    const int fl = (l31 >> 1) & 7;   // the reader's rows are 32 k + l31: (row >> 1) & 7 is the lane's
    int roff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) roff[g] = l31 * SM_BKC + 4 * ((2 * g + h) ^ fl);
    struct Frag {
        f32x4 b;
        f32x4 a[NT];
    };
    Frag f[2];
    auto read_frag = [&](Frag& fr, const float* base, int g) {
        fr.b = *reinterpret_cast<const f32x4*>(base + roff[g]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            fr.a[t] = *reinterpret_cast<const f32x4*>(base + (SM_TOK + 32 * t) * SM_BKC + roff[g]);
    };
    constexpr int NDMA = 4 + NWV;   // DMA wave-instructions per chunk

    f32x16 acc[NT];
    float best = __int_as_float(0x7F800000);
    int best_code = (tile_begin * CODE_TILE + wave * WROWS);

    // One chunk: the DMAs of the chunk two ahead (pointers wp, column koff) go to the idle stage; MFMAs on st0; before
    // the last k-group, wait until only this body's DMAs are in flight (the next chunk has landed) and read its first
    // fragments.
    // The issue order below is the program order: every step ends in sched_barrier(0), which nothing may cross (left to
    // itself hipcc clumps the twelve DMAs and the fragment reads in front of the chunk's MFMAs; sched_group_barrier
    // pipelines were not honoured around the M0 writes of the DMAs).
#define T5VQ_SB() __builtin_amdgcn_sched_barrier(0)
    // The DMA is an asm statement: hipcc's wait-count pass counts a global_load_lds builtin on the LDS counter as well and
    // then waits lgkmcnt(0) -- for the fragment reads just issued too -- in front of every MFMA group.  The statement's
    // VMEM operation is therefore unknown to that pass: every wait for DMA data below is written by hand (vmcnt counts
    // in order, so the compiler's own waits for ITS loads only become stricter by the extra operations in flight).
    auto issue_one = [&](float* stage, const gptr_t (&wp)[NWV], int koff, int q) {
        const gptr_t src = q < 4 ? xp[q] + koff : wp[q - 4] + koff;
        const unsigned dst = static_cast<unsigned>(reinterpret_cast<size_t>(
            (lds_ptr_t)(stage + (q < 4 ? 8 * q : SM_TOK + 8 * (q - 4)) * SM_BKC)));
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(dst), "v"(src) : "memory");   // (M0 is reserved: nothing else in this kernel uses it)
    };
    // prologue: chunks 0 and 1 of the first tile on their way; chunk 0 landed (a wave reads only what it fetched itself:
    // its own vmcnt wait orders the DMA's LDS writes before its ds_reads), its first fragments read
#pragma unroll
    for (int q = 0; q < NDMA; ++q) issue_one(st0, wpc, 0, q);
#pragma unroll
    for (int q = 0; q < NDMA; ++q) issue_one(st1, wpc, SM_BKC, q);
    // |x|^2 of this wave's 8 token rows, row_sqnorm_wave's way (lane l takes float4 #l, #l + 64, ...): requested now,
    // right behind the first two chunks, summed after the first code tile -- bit-identical to t5vq_row_sqnorms_f32
    f32x4 xr[8][NCH];
    if (!a.xx) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int t = tok_base + 8 * wave + i;
            if (t > N - 1) t = N - 1;
            const f32x4* row = reinterpret_cast<const f32x4*>(a.x + static_cast<size_t>(t) * D);
#pragma unroll
            for (int j = 0; j < NCH; ++j) xr[i][j] = row[j * 64 + lane];
        }
        T5VQ_SB();
        __builtin_amdgcn_s_waitcnt(vmcnt_only(NDMA + 8 * NCH));   // chunk 0 landed; chunk 1 and the rows may be in flight
    } else {
        __builtin_amdgcn_s_waitcnt(vmcnt_only(NDMA));
    }
    T5VQ_SB();
    read_frag(f[0], st0, 0);
#ifdef T5VQ_SMALL_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    T5VQ_STAMP(1);
#endif
    auto body = [&](const gptr_t (&wp)[NWV], int koff, auto fetch_t) {
        constexpr bool fetch = decltype(fetch_t)::value;
        constexpr int M = 4 * NT;                                    // MFMAs per k-group
        static_assert(NDMA + 1 <= 3 * M, "one DMA behind each MFMA from the chunk's second on, all before the last group");
        read_frag(f[1], st0, 1);
        T5VQ_SB();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[g & 1].a[t][j], f[g & 1].b[j], acc[t], 0, 0, 0);
                    const int m = j * NT + t;                        // position inside the group
                    const int q = g * M + m - 1;                      // DMA q goes behind the chunk's MFMA q + 1
#if T5VQ_SMALL_ABL == 0
                    if (fetch && q >= 0 && q < NDMA) issue_one(st2, wp, koff, q);
#elif T5VQ_SMALL_ABL == 2   // timing only: every chunk fetches the same columns (L1 / L2 hits, same issue and LDS traffic)
                    if (fetch && q >= 0 && q < NDMA) issue_one(st2, wp, 0, q);
#elif T5VQ_SMALL_ABL == 4   // timing only: the token rows are not fetched
                    if (fetch && q >= 4 && q < NDMA) issue_one(st2, wp, koff, q);
#endif
                    if (g == 3 && m == M / 2 - 1) {
                        // the next chunk has landed once only this body's DMAs are in flight (a wave reads only what it
                        // fetched itself: its own vmcnt wait orders the DMA's LDS writes before its ds_reads);
                        // s_waitcnt vmcnt(n) alone: expcnt 7, lgkmcnt 15
                        T5VQ_SB();
#if T5VQ_SMALL_ABL == 0 || T5VQ_SMALL_ABL == 2
                        __builtin_amdgcn_s_waitcnt(vmcnt_only(fetch ? NDMA : 0));
#elif T5VQ_SMALL_ABL == 4
                        __builtin_amdgcn_s_waitcnt(vmcnt_only(fetch ? NDMA - 4 : 0));
#else
                        __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
#endif
                        T5VQ_SB();
                        read_frag(f[0], st1, 0);
                    }
                    T5VQ_SB();
                }
            }
            // the fragment set this group used is free once its MFMAs are issued: the group after next reads into it
            if (g < 2) {
                read_frag(f[g & 1], st0, g + 2);
                T5VQ_SB();
            }
        }
        float* t0 = st0;
        st0 = st1;
        st1 = st2;
        st2 = t0;
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
        const bool more = ct + 1 < n_tiles;
        set_wptrs(wpn, more ? ct + 1 : ct);
        f32x4 e4[NT][4];
        load_ee(e4, ct);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // chunks 0 .. n_chunks - 3 fetch a chunk of this tile; the last two fetch chunks 0 and 1 of the next tile, or
        // nothing on the last tile
        for (int kc = 0; kc + 2 < n_chunks; ++kc) body(wpc, (kc + 2) * SM_BKC, std::true_type());
        if (more) {
            body(wpn, 0, std::true_type());
            body(wpn, SM_BKC, std::true_type());
        } else {
            body(wpn, 0, std::false_type());
            body(wpn, SM_BKC, std::false_type());
        }
#pragma unroll
        for (int j = 0; j < NWV; ++j) wpc[j] = wpn[j];
        if (ct == n_tiles - 1) T5VQ_STAMP(2);

        if (ct == 0 && !a.xx) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float sacc = 0.f;
#pragma unroll
                for (int j = 0; j < NCH; ++j) sacc = sq4(xr[i][j], sacc);
                sacc = wave_sum(sacc);
                if (lane == 0) s_xx[8 * wave + i] = sacc;
            }
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
                    const float sxe = xv + e4[t][g4][c];
                    const float d = fmaf(-2.0f, acc[t][g4 * 4 + c], sxe);
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
    // tokens of the tile (positions g_lo .. g_lo + g_cnt - 1) this workgroup finishes: all 32, or -- two code splits that
    // meet within a bounded wait -- its own half
    int g_lo = 0, g_cnt = SM_TOK;
    if (S > 1) {
        if (mine)
            __hip_atomic_store((gu64*)(a.slots + static_cast<size_t>(split) * N + my_tok),
                               static_cast<unsigned long long>(key), T5VQ_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the storing wave drains before it signals
        if (tid == 0) {
            gu32* ticket = (gu32*)(a.tile_ticket + tile);
            const unsigned old = __hip_atomic_fetch_add(ticket, 1u, T5VQ_RLX_AGENT);
            int role;   // 0 leave, 1 finish the whole tile, 2 finish my half
            if (S == 2) {
                // Two splits share the gather when they meet.  The first to arrive polls the ticket a bounded number of
                // times; if its partner shows up it takes its half, else it marks the ticket (+16) and leaves, and the
                // partner, seeing the mark, finishes the whole tile.  Whoever touches the word last resets it.  No
                // workgroup waits for another without a bound, so nothing depends on both being resident at once.
                if (old == 0) {
                    unsigned v = 1;
                    for (int spins = 0; spins < 12; ++spins) {
                        v = __hip_atomic_load(ticket, T5VQ_RLX_AGENT);
                        if (v >= 2) break;
                        __builtin_amdgcn_s_sleep(4);
                    }
                    if (v >= 2) {
                        role = 2;
                    } else {
                        const unsigned old2 = __hip_atomic_fetch_add(ticket, 16u, T5VQ_RLX_AGENT);
                        role = old2 >= 2 ? 2 : 0;   // the partner arrived after the last poll: it does its half only
                    }
                    if (role == 2) __hip_atomic_store(ticket, 0u, T5VQ_RLX_AGENT);
                } else {                  // second: 1 = the partner is (or was) polling, 17 = it left
                    role = old == 1 ? 2 : 1;
                    if (role == 1) __hip_atomic_store(ticket, 0u, T5VQ_RLX_AGENT);
                }
            } else {
                role = old == static_cast<unsigned>(S - 1) ? 1 : 0;
                if (role) __hip_atomic_store(ticket, 0u, T5VQ_RLX_AGENT);   // reset for the next launch
            }
            s_misc[0] = role;
        }
        __syncthreads();
        const int role = s_misc[0];
        if (!role) return;
        if (mine) {
            for (int s = 0; s < S; ++s) {
                if (s == split) continue;
                const long long o = static_cast<long long>(
                    __hip_atomic_load((gu64*)(a.slots + static_cast<size_t>(s) * N + my_tok), T5VQ_RLX_AGENT));
                key = o < key ? o : key;
            }
        }
        if (role == 2) {
            g_cnt = SM_TOK / 2;
            g_lo = split * g_cnt;
        }
    }
    if (tid < SM_TOK) {
        int code = -1;
        if (mine && tid >= g_lo && tid < g_lo + g_cnt) {
            code = key_index(key);
            code = code < 0 ? 0 : (code > K - 1 ? K - 1 : code);
            if (a.idx_out) a.idx_out[my_tok] = code;
        }
        s_code[tid] = code;   // -1: not this workgroup's to finish
    }
    __syncthreads();
    T5VQ_STAMP(4);

    // ---- gather + straight-through + loss terms: wave w takes cnt consecutive tokens, all rows requested at once.
    // What another workgroup waits for (loss terms, codes) leaves first and is signalled before the bulk stores go out.
    const int cnt = g_cnt / 4;   // 8 or 4 tokens per wave
    f32x4 xq[8][NCH], qq[8][NCH];
    int codes[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= cnt) break;
        const int pos = g_lo + cnt * wave + i;
        const int t = tok_base + pos;
        codes[i] = s_code[pos];
        const int tc = t < N ? t : N - 1;
        const int cc_ = codes[i] < 0 ? 0 : codes[i];
        const f32x4* xrow = reinterpret_cast<const f32x4*>(a.x + static_cast<size_t>(tc) * D);
        const f32x4* wr = reinterpret_cast<const f32x4*>(a.w + static_cast<size_t>(cc_) * D);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            xq[i][j] = xrow[j * 64 + lane];
            qq[i][j] = wr[j * 64 + lane];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= cnt) break;
        const int t = tok_base + g_lo + cnt * wave + i;
        if (t >= N) continue;   // wave-uniform
        float lacc = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const f32x4 df = qq[i][j] - xq[i][j];
            lacc = fmaf(df.x, df.x, lacc);
            lacc = fmaf(df.y, df.y, lacc);
            lacc = fmaf(df.z, df.z, lacc);
            lacc = fmaf(df.w, df.w, lacc);
        }
        if (a.loss_terms) {
            lacc = wave_sum(lacc);
            if (lane == 0) {
                __hip_atomic_store((gu32*)(a.loss_terms + t), __float_as_uint(lacc), T5VQ_RLX_AGENT);
                if (a.scalars) __hip_atomic_store((gu32*)(a.codes + t), static_cast<unsigned>(codes[i]), T5VQ_RLX_AGENT);
            }
        }
    }
    int last = 0;
    if (a.scalars) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its terms and codes have left
        __syncthreads();
        if (tid == 0) {
            // tokens this workgroup finished; the launch is complete when all N are (a workgroup whose half lies beyond
            // the batch adds nothing and cannot be last)
            int n_here = N - tok_base - g_lo;
            n_here = n_here < 0 ? 0 : (n_here > g_cnt ? g_cnt : n_here);
            int l = 0;
            if (n_here > 0) {
                const unsigned old =
                    __hip_atomic_fetch_add((gu32*)a.done_ticket, static_cast<unsigned>(n_here), T5VQ_RLX_AGENT);
                l = old + static_cast<unsigned>(n_here) == static_cast<unsigned>(N);
            }
            if (l) __hip_atomic_store((gu32*)a.done_ticket, 0u, T5VQ_RLX_AGENT);
            s_misc[2] = l;
        }
        __syncthreads();
        last = s_misc[2];
    }
    if (last) {
        // ---- the call's scalars, by the workgroup that signalled last (before its own bulk stores: the sc1 loads below
        // would otherwise wait behind them): every loss term and every code was stored sc1 and drained before its
        // workgroup's signal, and is read back with sc1 loads only (no acquire, no counts in global memory)
        T5VQ_STAMP(6);
        __syncthreads();
        int* hist = reinterpret_cast<int*>(lds);           // [K] in the staging area (the host checks that it fits)
        for (int k = tid; k < K; k += 256) hist[k] = 0;
        __syncthreads();
        {
            const long long n4 = N >> 2;
            for (long long q0 = tid; q0 < n4; q0 += 1024) {
                f32x4 u[4];
                const float* cp = reinterpret_cast<const float*>(a.codes);
                const bool k1 = q0 + 256 < n4, k2 = q0 + 512 < n4, k3 = q0 + 768 < n4;
                load4_sc1(cp + 4 * q0, cp + 4 * (q0 + 256), cp + 4 * (q0 + 512), cp + 4 * (q0 + 768), true, k1, k2, k3, u);
                const bool have[4] = {true, k1, k2, k3};
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (!have[m]) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(hist + __float_as_int(u[m][e]), 1);
                }
            }
            const int tl = static_cast<int>(n4 << 2) + tid;
            if (tl < N) atomicAdd(hist + static_cast<int>(__hip_atomic_load((const gu32*)a.codes + tl, T5VQ_RLX_AGENT)), 1);
        }
        __syncthreads();
        const double ent = entropy_as_1024(hist, K, a.n_tokens, s_red);
        const double lsum = a.loss_terms ? sum_terms_as_1024(a.loss_terms, N, s_red) : 0.0;
        if (a.counts)
            for (int k = tid; k < K; k += 256) a.counts[k] = static_cast<float>(hist[k]);
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
    // the bulk of the output
    float* trow = wl;   // the wave's staging area is free now: transposition row for the embed_sum flush
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= cnt) break;
        const int t = tok_base + g_lo + cnt * wave + i;
        if (t >= N) continue;
        const size_t rowoff = static_cast<size_t>(t) * D;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c4 = j * 64 + lane;
            const f32x4 df = qq[i][j] - xq[i][j];
            if (a.out_ste) reinterpret_cast<f32x4*>(a.out_ste + rowoff)[c4] = xq[i][j] + df;
            if (a.out_q) reinterpret_cast<f32x4*>(a.out_q + rowoff)[c4] = qq[i][j];
            if (a.out_diff) reinterpret_cast<f32x4*>(a.out_diff + rowoff)[c4] = df;
        }
        if (a.embed_sum) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) reinterpret_cast<f32x4*>(trow)[j * 64 + lane] = xq[i][j];
            float* dst = a.embed_sum + static_cast<size_t>(codes[i]) * D;
#pragma unroll
            for (int mcol = 0; mcol < NCH * 4; ++mcol) atomicAdd(dst + mcol * 64 + lane, trow[mcol * 64 + lane]);
        }
    }
    // counts (a call without fused scalars): one atomic per (workgroup, code) -- the tokens are grouped in LDS by wave 0
    if (a.counts && !a.scalars && wave == 0) {
        const int mine_c = lane < SM_TOK ? s_code[lane] : -1;
        unsigned long long m = 0;
        for (int t = 0; t < SM_TOK; ++t)
            if (s_code[t] == mine_c) m |= 1ull << t;
        if (mine_c >= 0 && (__ffsll(static_cast<long long>(m)) - 1) == lane)
            atomicAdd(a.counts + mine_c, static_cast<float>(__popcll(m)));
    }
    T5VQ_STAMP(5);
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
    return align_up(STATE_TICKET_BYTES, 256);   // the tickets only (K: room to grow without touching callers)
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
    T5VQ_I_REQUIRE(!scalars || (K % 4 == 0 && K <= 16384), "scalars in the same launch need K % 4 == 0 and K <= 16384");
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
    a.counts = counts;
    if (scalars) {   // the counts come out of the last workgroup's LDS histogram: nothing to zero
        a.codes = reinterpret_cast<int*>(base + off_xx);
    } else if (counts) {
        T5VQ_I_CHECK_HIP(hipMemsetAsync(counts, 0, static_cast<size_t>(K) * 4, stream));
    }
    if (embed_sum) T5VQ_I_CHECK_HIP(hipMemsetAsync(embed_sum, 0, static_cast<size_t>(K) * D * 4, stream));
#ifdef T5VQ_SMALL_STAMPS
    a.stamps = g_stamps;
#endif
    return plan.nt == 2 ? launch_small_nch<2, 2>(a, plan, stream) : launch_small_nch<1, 2>(a, plan, stream);
}

#ifdef T5VQ_SMALL_STAMPS
void t5vq_small_set_stamps(void* p) { g_stamps = static_cast<long long*>(p); }
#endif

}  // extern "C"
