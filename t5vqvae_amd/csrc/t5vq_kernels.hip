// t5vq_kernels.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for the VQ bottleneck.
// See include/t5vq.h for the contract and the reference lines each entry point replaces.
//
// Kernel set
//   row_sqnorms        |x|^2, |e|^2                         HBM-bound
//   dist_mfma<..>      x.E^T on v_mfma_f32_32x32x2_f32, LDS-staged tiles, fused argmin epilogue   MFMA-bound
//   gather_stats<..>   W[idx] gather + STE + loss terms + in-LDS segment sum -> one atomic flush   HBM/atomic-bound
//   ema_cluster / ema_embed  EMA + Laplace smoothing                                                HBM-bound
//   ste_backward / scatter_rows / cosine_backward / one_hot / unpack_keys
//
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <type_traits>

#include "t5vq.h"
#include "t5vq_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* a = "", const char* b = "") {
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}

}  // namespace
// error hook shared with the other translation units of the library (t5vq_internal.h)
int t5vq_internal_fail(int code, const char* fmt, const char* a, const char* b) { return fail(code, fmt, a, b); }
namespace {

#define T5VQ_CHECK_HIP(expr)                                                         \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) return fail(T5VQ_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define T5VQ_REQUIRE(cond, msg)                                   \
    do {                                                          \
        if (!(cond)) return fail(T5VQ_ERR_INVALID, "%s (%s)", msg, #cond); \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ------------------------------------------------------------------------------------------------
// packed keys
// ------------------------------------------------------------------------------------------------
constexpr long long KEY_NONE = 0x7FFFFFFFFFFFFFFFLL;
constexpr int CODE_NONE = 0x7FFFFFFF;  // 'no admissible code seen yet' (ranged mode)

__device__ __forceinline__ long long pack_key(float d, int idx) {
    int b = __float_as_int(d);
    int s = b >= 0 ? b : (b ^ 0x7FFFFFFF);  // signed, order-preserving image of the float
    return (static_cast<long long>(s) << 32) | static_cast<unsigned int>(idx);
}
__device__ __forceinline__ float key_dist(long long key) {
    int s = static_cast<int>(key >> 32);
    return __int_as_float(s >= 0 ? s : (s ^ 0x7FFFFFFF));
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

// |row|^2 by one wave: lane l takes float4 #l, #l+64, ... in a k-ordered fma chain, then the xor butterfly.  THE
// definition of a row norm in this library: every kernel that produces |x|^2 or |e|^2 (row_sqnorms_kernel, the
// self-contained distance kernel's prologue, ema_embed_kernel, filter_prep_kernel, rmsnorm_sqnorms_kernel) uses this
// lane layout and order, so their results are bit-identical and interchangeable.
__device__ __forceinline__ float sq4(const f32x4 v, float s) {
    s = fmaf(v.x, v.x, s);
    s = fmaf(v.y, v.y, s);
    s = fmaf(v.z, v.z, s);
    s = fmaf(v.w, v.w, s);
    return s;
}
__device__ __forceinline__ float row_sqnorm_wave(const float* __restrict__ row, int D, int lane) {
    const f32x4* p = reinterpret_cast<const f32x4*>(row);
    const int n4 = D >> 2;
    float s = 0.f;
    // four loads in flight per lane (a 768-float row is three of them): pieces past the row's end are zeros, and adding
    // 0 * 0 to a non-negative sum leaves its bits alone, so the chain is the plain one, float4 by float4 in order
    for (int base = lane; base < n4; base += 256) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = base + 64 * j;
            v[j] = i < n4 ? p[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) s = sq4(v[j], s);
    }
    return wave_sum(s);
}

// ------------------------------------------------------------------------------------------------
// K0: row squared norms (one wave per row, two matrices per launch)
// ------------------------------------------------------------------------------------------------
// Buffers a composite forward needs in a known state before its later kernels run; prepared by extra
// workgroups of the row-norm launch instead of separate fill / memset launches.
struct PrepArgs {
    float* zero_a;       // counts
    long long n_a;
    float* zero_b;       // embed_sum
    long long n_b;
    float* zero_c;       // the counting sort's cursors (int32 words)
    long long n_c;
    long long* keys;     // filled with KEY_NONE (code splits merge into it with atomicMin)
    long long n_keys;
    int blocks;          // workgroups appended to the grid for this
};

__device__ __forceinline__ void zero_floats(float* p, long long n, long long gtid, long long gstride) {
    if (!p || n <= 0) return;
    long long head = ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) >> 2;  // floats up to 16-byte alignment
    if (head > n) head = n;
    const long long body4 = (n - head) >> 2;
    f32x4* b = reinterpret_cast<f32x4*>(p + head);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long long i = gtid; i < body4; i += gstride) b[i] = z;
    if (gtid < head) p[gtid] = 0.f;
    const long long tail0 = head + (body4 << 2);
    if (gtid < n - tail0) p[tail0 + gtid] = 0.f;
}

__global__ __launch_bounds__(256) void row_sqnorms_kernel(const float* __restrict__ x, long long n_x,
                                                          float* __restrict__ xx, const float* __restrict__ w,
                                                          long long n_w, long long n_w_pad, float* __restrict__ ee,
                                                          int D, long long blocks_x, long long blocks_rows,
                                                          PrepArgs prep) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (static_cast<long long>(blockIdx.x) >= blocks_rows) {  // preparation tail
        const long long gtid = (static_cast<long long>(blockIdx.x) - blocks_rows) * 256 + threadIdx.x;
        const long long gstride = static_cast<long long>(prep.blocks) * 256;
        zero_floats(prep.zero_a, prep.n_a, gtid, gstride);
        zero_floats(prep.zero_b, prep.n_b, gtid, gstride);
        zero_floats(prep.zero_c, prep.n_c, gtid, gstride);
        if (prep.keys)
            for (long long i = gtid; i < prep.n_keys; i += gstride) prep.keys[i] = KEY_NONE;
        return;
    }
    const float* src;
    float* dst;
    long long row, n_rows, n_out;
    if (static_cast<long long>(blockIdx.x) < blocks_x) {
        src = x; dst = xx; n_rows = n_x; n_out = n_x;
        row = static_cast<long long>(blockIdx.x) * 4 + wave;
    } else {
        src = w; dst = ee; n_rows = n_w; n_out = n_w_pad;
        row = (static_cast<long long>(blockIdx.x) - blocks_x) * 4 + wave;
    }
    if (row >= n_out) return;
    if (row >= n_rows) {
        if (lane == 0) dst[row] = __int_as_float(0x7F800000);  // +inf: padded codes never win the argmin
        return;
    }
    const float s = row_sqnorm_wave(src + row * D, D, lane);
    if (lane == 0) dst[row] = s;
}

// ------------------------------------------------------------------------------------------------
// K1: fused distance + argmin on fp32 MFMA
//
// Workgroup = 256 threads = 4 waves; tile = 128 tokens x 128 codes; K-chunk (of D) = 32.
// C[code][token] = sum_k W[code][k] * X[token][k]:  A operand = codebook rows, B operand = token rows, so in
// the 32x32 accumulator the TOKEN sits on the lane (col = lane & 31) and 16 CODES sit in the registers
// (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).  The per-token argmin is then a register-local scan; only
// the two lane halves are merged at the very end (one shuffle).  Wave w owns tokens [32w, 32w+32) of the
// tile and all 128 codes: 4 independent accumulators.
// LDS: double-buffered [128][36] fp32 images of the X and W chunks (row stride 144 B = 9 x 16 B, so the 16
// lanes of every ds_read_b128 group hit 16 distinct 16-byte slots: conflict-free), filled through
// registers: global loads for chunk t+1 are issued before the MFMAs of chunk t and written after them.
// ------------------------------------------------------------------------------------------------
constexpr int TOK_TILE = 128;

// Tile configuration: NT = 32-code sub-tiles per workgroup tile (code tile = 32 NT), BKC = floats of D per
// LDS stage, PF = register double-buffering of the LDS fragments, MINW = waves per SIMD asked of the
// register allocator (= workgroups per CU here).
template <int NT_, int BKC_, bool PF_, int MINW_, int ABL_ = 0>
struct DistCfg {
    static constexpr int ABL = ABL_;  // timing-only ablations (tuning builds): 1 no staging, 2 +no barrier, 3 +no LDS reads
    static constexpr int NT = NT_;
    static constexpr int BKC = BKC_;
    static constexpr bool PF = PF_;
    static constexpr int MINW = MINW_;
    static constexpr int CODE_TILE = 32 * NT_;
    static constexpr int ROW = BKC_ + 4;                   // LDS row stride in floats: (BKC+4)*4 B = odd x 16 B
    static constexpr int X_IMG = TOK_TILE * ROW;           // floats
    static constexpr int W_IMG = CODE_TILE * ROW;
    static constexpr int STAGE = X_IMG + W_IMG;
    static constexpr int LDS_BYTES = 2 * STAGE * 4;
    static constexpr int F4_PER_ROW = BKC_ / 4;
    static constexpr int XV = TOK_TILE * F4_PER_ROW / 256;   // float4 per thread per stage
    static constexpr int WV = CODE_TILE * F4_PER_ROW / 256;
    static constexpr int ROWS_PER_PASS = 256 / F4_PER_ROW;   // rows covered by one float4 per thread
    static constexpr int G = BKC_ / 8;                       // k-groups of 8 per stage
};

template <class C>
struct Stage {
    f32x4 xv[C::XV];
    f32x4 wv[C::WV];
};
template <class C>
struct Frag {
    f32x4 b;
    f32x4 a[C::NT];
};

// BOUNDED (top-k passes): `range` is reinterpreted as one packed key per token and only (distance, code) pairs that
// order strictly AFTER that key are admissible -- pass j of a top-k search excludes the j best found so far.
// LISTED (decision scan of the filtered search): only the tokens named in tok_list[0 .. *n_listed) are searched.  The
// host does not know the list's length: the grid is a fixed few workgroups per CU that stride over the (token tile,
// split) pairs the list needs (a grid sized for all N tokens spent 13 us dispatching workgroups that return at once).
template <class C, bool RANGED, bool WRITE_DIST, bool BOUNDED = false, bool LISTED = false>
__global__ __launch_bounds__(256, C::MINW) void dist_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ xx,
    const float* __restrict__ ee, const long long* __restrict__ range, int N_all, int K, int D, int n_splits,
    int tiles_per_split, long long code_offset, int atomic_merge, long long* __restrict__ keys,
    float* __restrict__ dist, const int* __restrict__ tok_list = nullptr, const int* __restrict__ n_listed = nullptr) {
    int N = N_all;
    if (LISTED) {
        N = *n_listed;
        if (N > N_all) N = N_all;
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = C::NT, BKC = C::BKC, ROW = C::ROW, CODE_TILE = C::CODE_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // Blocks b and b + 8 share an XCD (round-robin dispatch; speed only, never correctness): the n_splits
    // workgroups of one token tile are given ids 8 apart so that their X tile is fetched into ONE L2.
    // One pass for a grid that covers every (token tile, split); LISTED: virtual block ids vb, vb + gridDim, ...
    // (`continue` in a do-while jumps to its condition: the early exits below end the pass, not the workgroup)
    unsigned vb = blockIdx.x;
    const unsigned vb_total =
        LISTED ? static_cast<unsigned>((N + 8 * TOK_TILE - 1) / (8 * TOK_TILE)) * 8u * static_cast<unsigned>(n_splits) : 0u;
    bool again = false;
    do {
    if (again) __syncthreads();   // the previous pass's LDS reads are done before this pass restages
    again = true;
    const int grp = vb / (8 * n_splits);
    int split = (vb / 8) % n_splits;
    if (RANGED && !LISTED) split = (split + grp) % n_splits;   // as in dist_mfma_v2_kernel
    const int tok_base = (grp * 8 + (vb & 7)) * TOK_TILE;
    if (tok_base >= N) continue;
    const int n_code_tiles = (K + CODE_TILE - 1) / CODE_TILE;
    const int tile_begin = split * tiles_per_split;
    int tile_end = tile_begin + tiles_per_split;
    if (tile_end > n_code_tiles) tile_end = n_code_tiles;
    const int n_tiles = tile_end - tile_begin;
    if (n_tiles <= 0) continue;
    const int n_chunks = (D + BKC - 1) / BKC;
    const int total = n_tiles * n_chunks;

    // staging assignment: float4 #f = tid + 256 j  ->  row f / F4_PER_ROW, 16-byte column f % F4_PER_ROW
    const int s_row = tid / C::F4_PER_ROW;             // + ROWS_PER_PASS * j
    const int s_col = (tid % C::F4_PER_ROW) * 4;       // float offset inside the chunk
    const float* xrow[C::XV];
#pragma unroll
    for (int j = 0; j < C::XV; ++j) {
        int t = tok_base + s_row + C::ROWS_PER_PASS * j;
        if (t > N - 1) t = N - 1;
        if (LISTED) t = tok_list[t];
        xrow[j] = x + static_cast<size_t>(t) * D + s_col;
    }

    auto load_stage = [&](Stage<C>& st, int ct, int kc) {
        const int col = kc * BKC + s_col;
        const bool in = col < D;  // D % 4 == 0, so a float4 is wholly in or out
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < C::XV; ++j) st.xv[j] = in ? *reinterpret_cast<const f32x4*>(xrow[j] + kc * BKC) : z;
#pragma unroll
        for (int j = 0; j < C::WV; ++j) {
            int c = (tile_begin + ct) * CODE_TILE + s_row + C::ROWS_PER_PASS * j;
            if (c > K - 1) c = K - 1;
            const float* wp = w + static_cast<size_t>(c) * D + s_col + kc * BKC;
            st.wv[j] = in ? *reinterpret_cast<const f32x4*>(wp) : z;
        }
    };
    auto store_stage = [&](const Stage<C>& st, int buf) {
        float* xs = lds + buf * C::STAGE;
        float* ws = xs + C::X_IMG;
#pragma unroll
        for (int j = 0; j < C::XV; ++j)
            *reinterpret_cast<f32x4*>(xs + (s_row + C::ROWS_PER_PASS * j) * ROW + s_col) = st.xv[j];
#pragma unroll
        for (int j = 0; j < C::WV; ++j)
            *reinterpret_cast<f32x4*>(ws + (s_row + C::ROWS_PER_PASS * j) * ROW + s_col) = st.wv[j];
    };

    // per-lane token data
    const int tok = tok_base + wave * 32 + l31;
    int tok_c = tok < N ? tok : N - 1;
    if (LISTED) tok_c = tok_list[tok_c];   // from here on tok_c is the token's row in x / xx / keys
    const float xv = xx[tok_c];
    long long r_start = 0, r_end = 0;
    if (RANGED) {
        r_start = range[2 * static_cast<size_t>(tok_c)];
        r_end = range[2 * static_cast<size_t>(tok_c) + 1];
    }
    int bound_s = 0, bound_c = 0;  // BOUNDED: order image of the excluded distance, excluded global code
    if (BOUNDED) {
        const long long bk = range[tok_c];
        bound_s = static_cast<int>(bk >> 32);
        bound_c = key_index(bk);
    }

    float best = __int_as_float(0x7F800000);
    int best_code = (RANGED || BOUNDED) ? CODE_NONE : tile_begin * CODE_TILE;  // local code id (before code_offset)

    Stage<C> st;
    load_stage(st, 0, 0);
    store_stage(st, 0);
    __syncthreads();

    int it = 0;
    for (int ct = 0; ct < n_tiles; ++ct) {
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        for (int kc = 0; kc < n_chunks; ++kc, ++it) {
            const bool has_next = it + 1 < total;
            const float* xs = lds + (it & 1) * C::STAGE + (wave * 32 + l31) * ROW + 4 * h;
            const float* ws = lds + (it & 1) * C::STAGE + C::X_IMG + l31 * ROW + 4 * h;
            auto read_frag = [&](Frag<C>& f, int g) {
                f.b = *reinterpret_cast<const f32x4*>(xs + 8 * g);
#pragma unroll
                for (int t = 0; t < NT; ++t) f.a[t] = *reinterpret_cast<const f32x4*>(ws + t * 32 * ROW + 8 * g);
            };
            auto mma = [&](const Frag<C>& f) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t][j], f.b[j], acc[t], 0, 0, 0);
            };
            if (C::PF) {
                Frag<C> f0, f1;
                read_frag(f0, 0);
                if (has_next) {
                    int nkc = kc + 1, nct = ct;
                    if (nkc == n_chunks) { nkc = 0; nct = ct + 1; }
                    load_stage(st, nct, nkc);
                }
#pragma unroll
                for (int g = 0; g < C::G; g += 2) {
                    if (g + 1 < C::G) read_frag(f1, g + 1);
                    mma(f0);
                    if (g + 2 < C::G) read_frag(f0, g + 2);
                    if (g + 1 < C::G) mma(f1);
                }
            } else {
                if (has_next && C::ABL < 1) {
                    int nkc = kc + 1, nct = ct;
                    if (nkc == n_chunks) { nkc = 0; nct = ct + 1; }
                    load_stage(st, nct, nkc);
                }
                if (C::ABL >= 3) {
                    Frag<C> f;
                    read_frag(f, 0);
#pragma unroll
                    for (int g = 0; g < C::G; ++g) mma(f);
                } else {
#pragma unroll
                    for (int g = 0; g < C::G; ++g) {
                        Frag<C> f;
                        read_frag(f, g);
                        mma(f);
                    }
                }
            }
            if (has_next && C::ABL < 1) store_stage(st, (it + 1) & 1);
            if (C::ABL < 2) __syncthreads();
        }

        // epilogue for this code tile: d = fl(fl(xx + ee) - 2 dot), running first-minimum per token
        const int code0 = (tile_begin + ct) * CODE_TILE + 4 * h;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int cb = code0 + t * 32 + 8 * g4;
                const f32x4 e4 = *reinterpret_cast<const f32x4*>(ee + cb);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int code = cb + c;
                    const float s = xv + e4[c];
                    float d = fmaf(-2.0f, acc[t][g4 * 4 + c], s);
                    if (WRITE_DIST) {
                        if (tok < N && code < K) dist[static_cast<size_t>(tok) * K + code] = d;
                    } else {
                        if (RANGED) {
                            const long long gc = code_offset + code;
                            const bool valid = gc >= r_start && gc < r_end && code < K;
                            if (valid && (d < best || best_code == CODE_NONE)) {  // first valid code always taken
                                best = d;
                                best_code = code;
                            }
                        } else if (BOUNDED) {
                            const int db = __float_as_int(d);
                            const int ds = db >= 0 ? db : (db ^ 0x7FFFFFFF);
                            const int gc = static_cast<int>(code_offset) + code;
                            const bool valid = code < K && (ds > bound_s || (ds == bound_s && gc > bound_c));
                            if (valid && (d < best || best_code == CODE_NONE)) {
                                best = d;
                                best_code = code;
                            }
                        } else {
                            if (d < best) {
                                best = d;
                                best_code = code;
                            }
                        }
                    }
                }
            }
        }
    }

    if (WRITE_DIST) continue;

    // merge the two lane halves (codes 4h + ...), lowest index wins ties
    const float ob = __shfl_xor(best, 32);
    const int oc = __shfl_xor(best_code, 32);
    if (oc != CODE_NONE && (best_code == CODE_NONE || ob < best || (ob == best && oc < best_code))) {
        best = ob;
        best_code = oc;
    }
    if (h == 0 && tok < N) {
        long long key = KEY_NONE;
        if (best_code != CODE_NONE) {
            long long gi = code_offset + best_code;
            if (RANGED) gi -= r_start;  // slice-relative, as the reference returns it
            key = pack_key(best, static_cast<int>(gi));
        }
        if (atomic_merge) {
            if (key != KEY_NONE) atomicMin(keys + tok_c, key);
        } else {
            keys[tok_c] = key;
        }
    }
    } while (LISTED && (vb += gridDim.x) < vb_total);
}

// ------------------------------------------------------------------------------------------------
// K1 (v2): the same tile and arithmetic as dist_mfma_kernel, restructured so that ONE wave keeps the matrix
// pipe fed by itself.  Measured on v1 (tools/tune_dist.py ablations, C3): 95 % of fp32 MFMA peak with the
// staging removed, 83 % with it -- the global loads and ds_writes were clumped before / after the MFMAs of a
// chunk, where the issuing wave starves the pipe.  Here the loop body is one basic block (no tail / has-next
// branches: D % BKC == 0 is required, the last iteration reloads a valid chunk into the idle buffer), source
// pointers are advanced instead of recomputed, LDS fragments are double-buffered in registers, and
// sched_group_barrier pins the order: every global load, ds_read and ds_write is issued in the 64-cycle
// shadow of an MFMA.
// ------------------------------------------------------------------------------------------------
#define T5VQ_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_MFMA = 0x8, SG_VMEM_RD = 0x20, SG_DS_RD = 0x100, SG_DS_WR = 0x200;

// D2: staging registers are double-buffered too, so a chunk's global loads are issued TWO chunks ahead of its
// use and have a whole chunk (~4k-8k cycles) to land before their ds_write (requires an even chunk count).
// R fragment reads spread over R0 MFMAs (sched_group_barrier needs literal counts: compile-time recursion)
template <int I, int R0, int R>
__device__ __forceinline__ void sgb_spread_reads() {
    if constexpr (I < R0) {
        T5VQ_SGB(SG_MFMA, 1);
        constexpr int c = R * (I + 1) / R0 - R * I / R0;
        if constexpr (c > 0) T5VQ_SGB(SG_DS_RD, c);
        sgb_spread_reads<I + 1, R0, R>();
    }
}

// SELF (small problems, where a launch costs as much as the work it carries): the kernel needs no launch before it.
//  * |x|^2 is taken from the X tile as it is staged for the first code tile, in row_sqnorm_wave's lane layout and
//    order -- K0's lane l holds float4 #(l + 64 j); the staging thread (row, c) sees float4 #(8 kc + c) in chunk kc,
//    i.e. K0 lane 8 (kc % 8) + c, so it keeps 8 accumulators per row (one per kc % 8, hence D % 256 == 0 and the
//    chunk loop unrolled by 8) and the xor butterfly runs first over those 8 (lane bits 5..3), then over the 8
//    threads of the row (lane bits 2..0): bit-identical to t5vq_row_sqnorms_f32.
//  * the code splits of a token tile do not merge into one prefilled key with atomicMin: split s writes its key to
//    keys[s * N + token] and the consumer (gather_stats_kernel) takes the minimum of the n_splits slots;
//  * the statistics buffers the following kernels add into are zeroed here, a slice per workgroup.
struct SelfArgs {
    float* zero[2];        // counts, embed_sum (16-byte aligned or null)
    long long nz[2];       // lengths in floats
};

template <class C, bool RANGED, bool D2, bool SELF = false>
__global__ __launch_bounds__(256, C::MINW) void dist_mfma_v2_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ xx,
    const float* __restrict__ ee, const long long* __restrict__ range, int N, int K, int D, int n_splits,
    int tiles_per_split, long long code_offset, int atomic_merge, long long* __restrict__ keys,
    SelfArgs self = SelfArgs{}, const int* __restrict__ order = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = C::NT, BKC = C::BKC, ROW = C::ROW, CODE_TILE = C::CODE_TILE, G = C::G;
    constexpr int NLD = C::XV + C::WV;
    // RANGED with `order` (a permutation of the token ids, t5vq_order_by_range): position p of the grid's token axis
    // stands for token order[p] -- its row of x, its norm, its range, its key.  Tokens of one role are neighbours in
    // `order`, so a token tile carries one or two ranges and skips the code tiles outside them, and nothing is moved
    // in memory for it (a row is 3 KB contiguous wherever it lies).
    auto token_at = [&](int p) { return (RANGED && order) ? order[p] : p; };
    static_assert(G >= 2 && NLD < 4 * NT, "interleave plan needs room");
    static_assert(!SELF || (D2 && !RANGED && BKC == 32 && C::F4_PER_ROW == 8), "SELF: plain search, staged two ahead");
    if (SELF) {   // every workgroup of the grid zeroes its slice of the statistics (before any early return)
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const long long n4 = self.nz[r] >> 2;   // lengths are multiples of 4 floats (checked on the host)
            const long long per = (n4 + gridDim.x - 1) / gridDim.x;
            const long long lo = per * blockIdx.x;
            const long long hi = lo + per < n4 ? lo + per : n4;
            for (long long i = lo + threadIdx.x; i < hi; i += 256) reinterpret_cast<f32x4*>(self.zero[r])[i] = z;
        }
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // Blocks b and b + 8 share an XCD (round-robin dispatch; speed only, never correctness): the n_splits
    // workgroups of one token tile are given ids 8 apart so that their X tile is fetched into ONE L2.
    const int grp = blockIdx.x / (8 * n_splits);
    int split = (blockIdx.x / 8) % n_splits;
    // RANGED: role-bucketed batches leave ONE split with work for long runs of token tiles.  Unswizzled, those
    // workgroups' ids inside an XCD would all share one residue mod n_splits and the dispatcher, which deals
    // consecutive workgroups round the XCD's shader arrays, would send them all to the same quarter of the CUs
    // (measured: one role of 1024 codes out of 8192 at N = 65536 took 3.0 ms against 0.72 ms of matrix work).
    // Rotating the split by the token group spreads them; which workgroup takes which split is free.
    if (RANGED) split = (split + grp) % n_splits;
    const int tok_base = (grp * 8 + (blockIdx.x & 7)) * TOK_TILE;
    if (tok_base >= N) return;
    const int n_code_tiles = (K + CODE_TILE - 1) / CODE_TILE;
    int tile_begin = split * tiles_per_split;
    int tile_end = tile_begin + tiles_per_split;
    if (tile_end > n_code_tiles) tile_end = n_code_tiles;
    if (!RANGED && tile_end <= tile_begin) {   // RANGED: the split's share is cut from the token tile's own range below
        if (SELF) {                            // an empty split still owns a key slot per token
            const int t = tok_base + (tid & 127);
            if (tid < TOK_TILE && t < N) keys[static_cast<size_t>(split) * N + t] = KEY_NONE;
        }
        return;
    }
    const int n_chunks = D / BKC;

    const int s_row = tid / C::F4_PER_ROW;
    const int s_col = (tid % C::F4_PER_ROW) * 4;
    // global address space stated explicitly: through a lambda parameter hipcc otherwise falls back to
    // flat loads, which also count on lgkmcnt and stall the LDS fragment waits behind HBM latency
    typedef const f32x4 __attribute__((address_space(1))) * gvec_t;
    typedef const float __attribute__((address_space(1))) * gptr_t;
    const gptr_t xg = (gptr_t)x;
    const gptr_t wg = (gptr_t)w;
    gptr_t xp[C::XV];
#pragma unroll
    for (int j = 0; j < C::XV; ++j) {
        int t = tok_base + s_row + C::ROWS_PER_PASS * j;
        if (t > N - 1) t = N - 1;
        xp[j] = xg + static_cast<size_t>(token_at(t)) * D + s_col;
    }
    gptr_t wpc[C::WV];
    gptr_t wpn[C::WV];
    // RANGED: workgroup-relative tile ct is code tile tile_begin + (ct + rot) % n_tiles (rot set below); plain: rot = 0
    int rot = 0, n_tiles_rot = 1 << 30;
    auto tile_at = [&](int ct) {
        int t = ct + rot;
        if (RANGED && t >= n_tiles_rot) t -= n_tiles_rot;
        return tile_begin + t;
    };
    auto set_wptrs = [&](gptr_t (&wp)[C::WV], int ct) {
#pragma unroll
        for (int j = 0; j < C::WV; ++j) {
            int c = tile_at(ct) * CODE_TILE + s_row + C::ROWS_PER_PASS * j;
            if (c > K - 1) c = K - 1;
            wp[j] = wg + static_cast<size_t>(c) * D + s_col;
        }
    };
    const int st_off = s_row * ROW + s_col;   // + ROWS_PER_PASS * ROW * j

    const int tok = tok_base + wave * 32 + l31;   // position on the token axis; tok_c = the token it stands for
    const int tok_c = token_at(tok < N ? tok : N - 1);
    float xv = (SELF && !xx) ? 0.f : xx[tok_c];   // SELF without given norms: set after the first code tile
    float xacc[SELF ? C::XV : 1][SELF ? 8 : 1];    // SELF: |x|^2 accumulators, [staged row j][chunk index mod 8]
    if (SELF) {
#pragma unroll
        for (int j = 0; j < C::XV; ++j)
#pragma unroll
            for (int m = 0; m < 8; ++m) xacc[j][m] = 0.f;
    }
    long long r_start = 0, r_end = 0;
    if (RANGED) {
        r_start = range[2 * static_cast<size_t>(tok_c)];
        r_end = range[2 * static_cast<size_t>(tok_c) + 1];
        // Only code tiles that intersect the union of this token tile's ranges are visited: with tokens
        // bucketed by role (vqvae.py does that before the call) a grouped codebook costs
        // 2 D sum_r N_r K_r flops instead of 2 D N K.
        long long lo = r_start - code_offset, hi = r_end - code_offset;
        lo = lo < 0 ? 0 : lo;
        hi = hi > K ? K : hi;
        int tlo = static_cast<int>(lo / CODE_TILE), thi = static_cast<int>((hi + CODE_TILE - 1) / CODE_TILE);
        if (hi <= lo) { tlo = 0x7FFFFFFF; thi = 0; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            tlo = min(tlo, __shfl_xor(tlo, o));
            thi = max(thi, __shfl_xor(thi, o));
        }
        int* red = reinterpret_cast<int*>(lds);
        if (lane == 0) { red[2 * wave] = tlo; red[2 * wave + 1] = thi; }
        __syncthreads();
        tlo = min(min(red[0], red[2]), min(red[4], red[6]));
        thi = max(max(red[1], red[3]), max(red[5], red[7]));
        __syncthreads();   // red is overwritten by the prologue below
        // The n_splits workgroups of this token tile divide ITS admissible code tiles [tlo, thi) among themselves
        // (not the whole codebook): with role-bucketed tokens every workgroup of the grid then has an equal share of
        // real work, whatever the role layout.  (Splitting the codebook statically left 7 of 8 workgroups idle at 8
        // roles and the busy ones unevenly spread over the CUs: 1.34 ms against 0.72 ms of matrix work.)
        const int per = (max(thi - tlo, 0) + n_splits - 1) / n_splits;
        tile_begin = tlo + split * per;
        tile_end = min(tile_begin + per, thi);
        if (tile_end <= tile_begin) {   // workgroup-uniform: nothing admissible for this workgroup
            if (!atomic_merge && h == 0 && tok < N) keys[tok_c] = KEY_NONE;
            return;
        }
    }
    const int n_tiles = tile_end - tile_begin;
    if (RANGED) {
        // Role-bucketed batches put tokens of ONE role into all the token tiles an XCD holds at a time; walking the
        // role's code tiles in the same order, 64 workgroups would ask one L2 channel for the same lines at the same
        // moment (measured: 8 roles of 1024 codes at N = 65536, 1.45 ms against 0.72 ms of matrix work).  Each token
        // tile therefore starts at its own code tile and wraps around; ties are resolved by code id, not by visiting
        // order (below), so the keys do not change.
        n_tiles_rot = n_tiles;
        rot = (tok_base / TOK_TILE) % n_tiles;
    }
    float best = __int_as_float(0x7F800000);
    int best_code = RANGED ? CODE_NONE : tile_begin * CODE_TILE;

    set_wptrs(wpc, 0);
    {   // prologue: chunk 0 of tile 0 into buffer 0
#pragma unroll
        for (int j = 0; j < C::XV; ++j) {
            const f32x4 v = *(gvec_t)(xp[j]);
            *reinterpret_cast<f32x4*>(lds + st_off + C::ROWS_PER_PASS * ROW * j) = v;
            if (SELF) xacc[j][0] = sq4(v, xacc[j][0]);
        }
#pragma unroll
        for (int j = 0; j < C::WV; ++j)
            *reinterpret_cast<f32x4*>(lds + C::X_IMG + st_off + C::ROWS_PER_PASS * ROW * j) =
                *(gvec_t)(wpc[j]);
    }
    Stage<C> stg[2];
    if (D2) {   // chunk 1 of tile 0 waits in registers
#pragma unroll
        for (int j = 0; j < C::XV; ++j) stg[1].xv[j] = *(gvec_t)(xp[j] + BKC);
#pragma unroll
        for (int j = 0; j < C::WV; ++j) stg[1].wv[j] = *(gvec_t)(wpc[j] + BKC);
    }
    __syncthreads();

    const int rd_x = (wave * 32 + l31) * ROW + 4 * h;
    const int rd_w = C::X_IMG + l31 * ROW + 4 * h;
    int buf = 0;
    f32x16 acc[NT];

    // One chunk: MFMAs on buffer `buf`, staging of the next chunk (source pointers wp, column koff) into buf^1.
    // P = parity of the chunk (D2 only): loads go to stg[P], the ds_writes drain stg[P ^ 1].
    using NoAcc = std::integral_constant<int, -1>;
    auto body = [&](const gptr_t (&wp)[C::WV], int koff, auto parity, auto accm) {
        constexpr int P = decltype(parity)::value;
        constexpr int AM = decltype(accm)::value;
        const float* xs = lds + buf * C::STAGE + rd_x;
        const float* ws = lds + buf * C::STAGE + rd_w;
        float* dst = lds + (buf ^ 1) * C::STAGE + st_off;
        Frag<C> f[2];
        Stage<C> st_local;
        Stage<C>& st = D2 ? stg[P] : st_local;          // filled by this chunk's loads
        Stage<C>& sw = D2 ? stg[P ^ 1] : st_local;      // written to LDS by this chunk
        auto read_frag = [&](Frag<C>& fr, int g) {
            fr.b = *reinterpret_cast<const f32x4*>(xs + 8 * g);
#pragma unroll
            for (int t = 0; t < NT; ++t) fr.a[t] = *reinterpret_cast<const f32x4*>(ws + t * 32 * ROW + 8 * g);
        };
        read_frag(f[0], 0);
        if (C::ABL != 5) {
#pragma unroll
            for (int j = 0; j < C::XV; ++j) st.xv[j] = *(gvec_t)(xp[j] + koff);
#pragma unroll
            for (int j = 0; j < C::WV; ++j) st.wv[j] = *(gvec_t)(wp[j] + koff);
        } else {   // timing-only ablation: no global loads
#pragma unroll
            for (int j = 0; j < C::XV; ++j) st.xv[j] = f[0].b;
#pragma unroll
            for (int j = 0; j < C::WV; ++j) st.wv[j] = f[0].a[0];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g + 1 < G) read_frag(f[(g + 1) & 1], g + 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[g & 1].a[t][j], f[g & 1].b[j], acc[t], 0, 0, 0);
        }
        if (C::ABL != 4) {
#pragma unroll
            for (int j = 0; j < C::XV; ++j) {
                *reinterpret_cast<f32x4*>(dst + C::ROWS_PER_PASS * ROW * j) = sw.xv[j];
                if constexpr (SELF && AM >= 0) xacc[j][AM] = sq4(sw.xv[j], xacc[j][AM]);
            }
#pragma unroll
            for (int j = 0; j < C::WV; ++j)
                *reinterpret_cast<f32x4*>(dst + C::X_IMG + C::ROWS_PER_PASS * ROW * j) = sw.wv[j];
        } else {   // timing-only ablation: no ds_writes, loads kept alive
#pragma unroll
            for (int j = 0; j < C::XV; ++j) asm volatile("" ::"v"(st.xv[j]));
#pragma unroll
            for (int j = 0; j < C::WV; ++j) asm volatile("" ::"v"(st.wv[j]));
        }

        // ---- issue order for the block above (M MFMAs and R fragment reads per k-group, NLD loads / writes)
        constexpr int M = 4 * NT, R = NT + 1, R0 = M - NLD, Q = M / NLD;
        T5VQ_SGB(SG_DS_RD, R);                            // fragments of k-group 0
#pragma unroll
        for (int i = 0; i < NLD; ++i) {                   // k-group 0: one global load per MFMA ...
            T5VQ_SGB(SG_MFMA, 1);
            T5VQ_SGB(SG_VMEM_RD, 1);
        }
        if constexpr (R <= R0) {                          // ... then the fragments of k-group 1
#pragma unroll
            for (int i = 0; i < R; ++i) {
                T5VQ_SGB(SG_MFMA, 1);
                T5VQ_SGB(SG_DS_RD, 1);
            }
            if constexpr (R0 - R > 0) T5VQ_SGB(SG_MFMA, R0 - R);
        } else {
            sgb_spread_reads<0, R0, R>();
        }
#pragma unroll
        for (int g = 1; g + 1 < G; ++g) {                 // middle k-groups: next fragments behind the MFMAs
#pragma unroll
            for (int i = 0; i < R; ++i) {
                T5VQ_SGB(SG_MFMA, 1);
                T5VQ_SGB(SG_DS_RD, 1);
            }
            T5VQ_SGB(SG_MFMA, M - R);
        }
        if (M - Q * NLD > 0) T5VQ_SGB(SG_MFMA, M - Q * NLD);   // last k-group: ds_writes at its end
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            T5VQ_SGB(SG_MFMA, Q);
            T5VQ_SGB(SG_DS_WR, 1);
        }
        __syncthreads();
        buf ^= 1;
    };

    for (int ct = 0; ct < n_tiles; ++ct) {
        set_wptrs(wpn, ct + 1 < n_tiles ? ct + 1 : ct);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        if (SELF && ct == 0) {
            // first code tile.  The body of MFMA chunk c writes chunk c + 1 of the X tile to LDS (loaded one body
            // earlier): those rows feed accumulator (c + 1) % 8; the tile's last body writes chunk 0 of the NEXT
            // pass over X and feeds nothing.  The loop keeps its two-bodies-per-trip shape (unrolling it by 8 to get
            // literal accumulator indices spilled 153 registers): the accumulators are a ring that is turned by two
            // places after every trip, so the bodies always feed places 1 and 2; n_chunks % 8 == 0 brings the ring
            // back to its starting position at the end of the tile.
            using A1 = std::integral_constant<int, 1>;
            using A2 = std::integral_constant<int, 2>;
            auto turn = [&]() {
#pragma unroll
                for (int j = 0; j < C::XV; ++j) {
                    const float t0 = xacc[j][0], t1 = xacc[j][1];
#pragma unroll
                    for (int m = 0; m < 6; ++m) xacc[j][m] = xacc[j][m + 2];
                    xacc[j][6] = t0;
                    xacc[j][7] = t1;
                }
            };
            for (int kc = 0; kc + 2 < n_chunks; kc += 2) {
                body(wpc, (kc + 2) * BKC, P0(), A1());
                body(wpc, (kc + 3) * BKC, P1(), A2());
                turn();
            }
            body(wpn, 0, P0(), A1());
            body(wpn, BKC, P1(), NoAcc());
            turn();
            if (!xx) {
                // the xor butterfly of wave_sum: lane bits 5..3 = the 8 accumulators of a thread, bits 2..0 = the 8
                // threads of a row (tid % 8); the row's sum goes through LDS to the lane that holds the token
                float* xs_ = lds + 2 * C::STAGE;   // 128 floats behind the two stages
#pragma unroll
                for (int j = 0; j < C::XV; ++j) {
                    const float n0 = xacc[j][0] + xacc[j][4], n1 = xacc[j][1] + xacc[j][5];
                    const float n2 = xacc[j][2] + xacc[j][6], n3 = xacc[j][3] + xacc[j][7];
                    const float p0 = n0 + n2, p1 = n1 + n3;
                    float q = p0 + p1;
                    q += __shfl_xor(q, 4);
                    q += __shfl_xor(q, 2);
                    q += __shfl_xor(q, 1);
                    if ((tid & 7) == 0) xs_[s_row + C::ROWS_PER_PASS * j] = q;
                }
                __syncthreads();
                xv = xs_[wave * 32 + l31];
            }
        } else if (D2) {
            for (int kc = 0; kc + 2 < n_chunks; kc += 2) {
                body(wpc, (kc + 2) * BKC, P0(), NoAcc());
                body(wpc, (kc + 3) * BKC, P1(), NoAcc());
            }
            body(wpn, 0, P0(), NoAcc());     // last two chunks: stage chunks 0 and 1 of the next tile
            body(wpn, BKC, P1(), NoAcc());   // (harmless reloads on the last tile)
        } else {
            for (int kc = 1; kc < n_chunks; ++kc) body(wpc, kc * BKC, P0(), NoAcc());
            body(wpn, 0, P0(), NoAcc());   // last chunk of the tile: stage chunk 0 of the next tile
        }
#pragma unroll
        for (int j = 0; j < C::WV; ++j) wpc[j] = wpn[j];

        const int code0 = tile_at(ct) * CODE_TILE + 4 * h;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int cb = code0 + t * 32 + 8 * g4;
                const f32x4 e4 = *reinterpret_cast<const f32x4*>(ee + cb);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int code = cb + c;
                    const float s = xv + e4[c];
                    const float d = fmaf(-2.0f, acc[t][g4 * 4 + c], s);
                    if (RANGED) {
                        const long long gc = code_offset + code;
                        const bool valid = gc >= r_start && gc < r_end && code < K;
                        // first admissible code always taken; equal distances: the lower code wins whatever the
                        // order in which the tiles are visited
                        if (valid && (d < best || best_code == CODE_NONE || (d == best && code < best_code))) {
                            best = d;
                            best_code = code;
                        }
                    } else {
                        if (d < best) {
                            best = d;
                            best_code = code;
                        }
                    }
                }
            }
        }
    }

    const float ob = __shfl_xor(best, 32);
    const int oc = __shfl_xor(best_code, 32);
    if (oc != CODE_NONE && (best_code == CODE_NONE || ob < best || (ob == best && oc < best_code))) {
        best = ob;
        best_code = oc;
    }
    if (h == 0 && tok < N) {
        long long key = KEY_NONE;
        if (best_code != CODE_NONE) {
            long long gi = code_offset + best_code;
            if (RANGED) gi -= r_start;
            key = pack_key(best, static_cast<int>(gi));
        }
        if (SELF) {
            keys[static_cast<size_t>(split) * N + tok] = key;   // one slot per (split, token): no prefill, no atomics
        } else if (atomic_merge) {
            if (key != KEY_NONE) atomicMin(keys + tok_c, key);
        } else {
            keys[tok_c] = key;
        }
    }
}

__global__ void fill_keys_kernel(long long* keys, long long n) {
    long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = KEY_NONE;
}

__global__ void unpack_keys_kernel(const long long* __restrict__ keys, long long n, long long* __restrict__ idx,
                                   float* __restrict__ dmin) {
    long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long k = keys[i];
    if (idx) idx[i] = key_index(k);
    if (dmin) dmin[i] = key_dist(k);
}

// keys[i] = min over k of pack(fl(fl(s * dl[i,k]) + fl(t * dr[i,k])), k): one wave per row of two distance matrices
__global__ __launch_bounds__(256) void blend_argmin_kernel(const float* __restrict__ dl, const float* __restrict__ dr,
                                                           float s, float t, int N, int K,
                                                           long long* __restrict__ keys) {
    const int lane = threadIdx.x & 63;
    const long long row = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* a = dl + row * K;
    const float* b = dr + row * K;
    long long best = KEY_NONE;
    for (int k = lane; k < K; k += 64) {
        const float d = __fadd_rn(__fmul_rn(s, a[k]), __fmul_rn(t, b[k]));
        const long long key = pack_key(d, k);
        best = key < best ? key : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long other = __shfl_xor(best, o);
        best = other < best ? other : best;
    }
    if (lane == 0) keys[row] = best;
}

// ------------------------------------------------------------------------------------------------
// K2+K3: gather + STE + loss + EMA statistics.  One workgroup = 64 tokens.
// Tokens of the workgroup that chose the same code form a group (64-bit member mask, built in LDS);
// a wave walks a group's members in ascending token order, summing their x rows in registers, and issues
// ONE flush of the group sum per (workgroup, code): the row is transposed through LDS so that every
// global_atomic_add_f32 wave-instruction covers 256 contiguous bytes.
// Lane layout inside a row: float4 #(j*64 + lane), j < NCH  (NCH = ceil(D / 256)).
// ------------------------------------------------------------------------------------------------
// Tokens ordered by code (counting sort; order inside a code is arrival order): with it a workgroup of the
// statistics kernel sees runs of equal codes and flushes one partial row per run instead of one per token --
// at C3 (8 tokens per code, spread over the batch) 189 -> 104 us, the gap between the atomic roof and the HBM roof.
// Whether that pays is decided ON THE DEVICE from the histogram pass: it also counts the (64-token block, code)
// groups the plain walk would flush; the scan compares that with distinct codes + N/64, the ordered walk's
// flushes, and the placement writes the identity when the difference would not cover the sort (clustered or
// collapsed batches, where neighbours already share codes).  Lanes of a wave that hold the same code are
// combined before any atomic, so a collapsed codebook does not serialise on one counter.
__device__ __forceinline__ unsigned long long same_code_lanes(int code, bool valid, int* n_distinct) {
    unsigned long long todo = __ballot(valid);
    unsigned long long mine = 0;
    int distinct = 0;
    while (todo) {
        const int leader = __ffsll(static_cast<long long>(todo)) - 1;
        const int c = __shfl(code, leader);
        const unsigned long long m = __ballot(valid && code == c);
        if (valid && code == c) mine = m;
        todo &= ~m;
        ++distinct;
    }
    *n_distinct = distinct;
    return mine;
}
// exclusive prefix of `local` over the threads of a workgroup (<= 1024): wave scans by shuffles, wave totals through LDS
__device__ __forceinline__ int block_exclusive_scan(int local, int* s_wave /*[16]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int i = 0; i < wave; ++i) before += s_wave[i];
    return before + incl - local;
}
// work layout: [0, K) counters / cursors, [K] groups of the plain walk, [K+1] 1 = order the tokens, 0 = identity
__global__ __launch_bounds__(256) void code_hist_kernel(const long long* __restrict__ keys, int N, int K,
                                                        int* __restrict__ work) {
    __shared__ int s_groups;
    if (threadIdx.x == 0) s_groups = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < N;
    int code = valid ? key_index(keys[i]) : 0;
    code = code < 0 ? 0 : (code > K - 1 ? K - 1 : code);
    int distinct;
    const unsigned long long m = same_code_lanes(code, valid, &distinct);
    if (valid && __ffsll(static_cast<long long>(m)) - 1 == lane) atomicAdd(work + code, __popcll(m));
    if (lane == 0) atomicAdd(&s_groups, distinct);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(work + K, s_groups);
}
// PER > 0: each thread owns PER consecutive counters, kept in registers between the two passes and moved as int4 where
// the pointer allows (K = 8192: two loads and two stores per thread instead of sixteen dependent round trips).
// PER == 0: any K, the counters are read twice.
template <int PER>
__global__ __launch_bounds__(1024) void code_scan_kernel(int* __restrict__ work, int N, int K) {  // in place: exclusive scan
    __shared__ int s_wave[16];
    __shared__ int s_used[16];
    const int per = PER > 0 ? PER : (K + 1023) / 1024;
    const int b = threadIdx.x * per;
    int local = 0, used = 0;
    int c[PER > 0 ? PER : 1];
    const bool vec = PER >= 4 && (reinterpret_cast<uintptr_t>(work) & 15u) == 0;
    if (PER > 0) {
#pragma unroll
        for (int j = 0; j < PER; j += 4) {
            if (PER >= 4 && vec && b + j + 3 < K) {
                const int4 v = *reinterpret_cast<const int4*>(work + b + j);
                c[j] = v.x;
                if (PER >= 4) { c[j + 1 < PER ? j + 1 : 0] = v.y; c[j + 2 < PER ? j + 2 : 0] = v.z; c[j + 3 < PER ? j + 3 : 0] = v.w; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j + e < PER) c[j + e] = b + j + e < K ? work[b + j + e] : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            local += c[j];
            used += c[j] > 0;
        }
    } else {
        for (int i = 0; i < per; ++i)
            if (b + i < K) {
                const int v = work[b + i];
                local += v;
                used += v > 0;
            }
    }
    used = static_cast<int>(wave_sum(static_cast<float>(used)));   // <= 64 * per: exact in fp32
    if ((threadIdx.x & 63) == 0) s_used[threadIdx.x >> 6] = used;
    int run = block_exclusive_scan(local, s_wave);
    if (PER > 0) {
#pragma unroll
        for (int j = 0; j < PER; j += 4) {
            int o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = run;
                if (j + e < PER) run += c[j + e];
            }
            if (PER >= 4 && vec && b + j + 3 < K) {
                *reinterpret_cast<int4*>(work + b + j) = int4{o[0], o[1], o[2], o[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j + e < PER && b + j + e < K) work[b + j + e] = o[e];
            }
        }
    } else {
        for (int i = 0; i < per; ++i)
            if (b + i < K) {
                const int v = work[b + i];
                work[b + i] = run;
                run += v;
            }
    }
    if (threadIdx.x == 0) {
        int codes_used = 0;
        for (int i = 0; i < 16; ++i) codes_used += s_used[i];
        const long long plain = work[K];
        const long long ordered = codes_used + (N + 63) / 64;
        work[K + 1] = (plain - ordered) * 4 > N ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void code_place_kernel(const long long* __restrict__ keys, int N, int K,
                                                         int* __restrict__ work, int* __restrict__ perm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < N;
    if (!work[K + 1]) {           // the same value for every thread of the grid
        if (valid) perm[i] = i;
        return;
    }
    int code = valid ? key_index(keys[i]) : 0;
    code = code < 0 ? 0 : (code > K - 1 ? K - 1 : code);
    int distinct;
    const unsigned long long m = same_code_lanes(code, valid, &distinct);
    const int leader = __ffsll(static_cast<long long>(m)) - 1;
    int base = 0;
    if (valid && leader == lane) base = atomicAdd(work + code, __popcll(m));
    base = __shfl(base, leader < 0 ? lane : leader);
    if (valid) perm[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
}

// ---- tokens ordered by their code range (disentangled codebooks): a counting sort over (start tile, end tile) buckets.
// Tokens of one role become neighbours, so a token tile of the search sees one or two ranges.  The granularity only
// has to be fine enough for that: ranges that share a bucket merely share a tile's visit list.
struct RangeBins {
    long long code_offset;   // global id of the first code of this codebook (slice)
    int K;                   // codes of this codebook (slice)
    int shift;               // log2 of the bucket granularity in codes
    int T;                   // buckets per axis: start in [0, T), end in [1, T]; bucket T (T + 1) = no admissible code
};
__host__ __device__ inline int range_bin_count(int T) { return T * (T + 1) + 1; }
__device__ __forceinline__ int range_bucket(const long long* __restrict__ range, int i, const RangeBins& rb) {
    long long a = range[2 * static_cast<size_t>(i)] - rb.code_offset;
    long long e = range[2 * static_cast<size_t>(i) + 1] - rb.code_offset;
    a = a < 0 ? 0 : (a > rb.K ? rb.K : a);
    e = e < 0 ? 0 : (e > rb.K ? rb.K : e);
    if (e <= a) return rb.T * (rb.T + 1);
    const int st = static_cast<int>(a >> rb.shift);                           // < T (a < K)
    const int et = static_cast<int>((e + (1LL << rb.shift) - 1) >> rb.shift); // 1 .. T
    return st * (rb.T + 1) + et;
}
// A role-ordered batch has a handful of distinct buckets, and every wave of the grid added to the same few words: the
// additions to one address serialise (about 10 ns each; 1024 waves on 8 buckets: 13 us per kernel at N = 65536).  The
// 16 waves of a workgroup therefore meet in a small LDS table first (bucket -> tokens of this workgroup), and one lane per
// table entry goes to memory.  A workgroup with more distinct buckets than the table holds (unordered ranges) sends the
// surplus straight to memory, as before.
constexpr int RANGE_WG = 1024, RANGE_SLOTS = 64;
struct RangeTable {
    int key[RANGE_SLOTS];    // bucket, -1 = free
    int count[RANGE_SLOTS];  // tokens of this workgroup in the bucket
    int base[RANGE_SLOTS];   // range_place_kernel: the workgroup's first position in the bucket
};
__device__ __forceinline__ void range_table_init(RangeTable& t) {
    if (threadIdx.x < RANGE_SLOTS) {
        t.key[threadIdx.x] = -1;
        t.count[threadIdx.x] = 0;
    }
    __syncthreads();
}
// slot of bucket b (claimed if new), or -1 when the table is full; *offset = tokens of the bucket before this wave's
__device__ __forceinline__ int range_table_add(RangeTable& t, int b, int n, int* offset) {
    int s = (b * 0x9E3779B1u) >> 26;   // 6 bits
    for (int probe = 0; probe < RANGE_SLOTS; ++probe, s = (s + 1) & (RANGE_SLOTS - 1)) {
        const int k = atomicCAS(&t.key[s], -1, b);
        if (k == -1 || k == b) {
            *offset = atomicAdd(&t.count[s], n);
            return s;
        }
    }
    return -1;
}
__global__ __launch_bounds__(RANGE_WG) void range_hist_kernel(const long long* __restrict__ range, int N, RangeBins rb,
                                                              int* __restrict__ work) {
    __shared__ RangeTable t;
    range_table_init(t);
    const int i = blockIdx.x * RANGE_WG + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < N;
    const int b = valid ? range_bucket(range, i, rb) : 0;
    int distinct;
    const unsigned long long m = same_code_lanes(b, valid, &distinct);
    if (valid && __ffsll(static_cast<long long>(m)) - 1 == lane) {
        int off;
        if (range_table_add(t, b, __popcll(m), &off) < 0) atomicAdd(work + b, __popcll(m));
    }
    __syncthreads();
    if (threadIdx.x < RANGE_SLOTS && t.key[threadIdx.x] >= 0) atomicAdd(work + t.key[threadIdx.x], t.count[threadIdx.x]);
}
__global__ __launch_bounds__(RANGE_WG) void range_place_kernel(const long long* __restrict__ range, int N, RangeBins rb,
                                                               int* __restrict__ work, int* __restrict__ order) {
    __shared__ RangeTable t;
    range_table_init(t);
    const int i = blockIdx.x * RANGE_WG + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < N;
    const int b = valid ? range_bucket(range, i, rb) : 0;
    int distinct;
    const unsigned long long m = same_code_lanes(b, valid, &distinct);
    const int leader = __ffsll(static_cast<long long>(m)) - 1;
    int slot = -1, base = 0;
    if (valid && leader == lane) {
        slot = range_table_add(t, b, __popcll(m), &base);             // base: offset inside the workgroup's share
        if (slot < 0) base = atomicAdd(work + b, __popcll(m));        // table full: straight to memory
    }
    __syncthreads();
    if (threadIdx.x < RANGE_SLOTS && t.key[threadIdx.x] >= 0)
        t.base[threadIdx.x] = atomicAdd(work + t.key[threadIdx.x], t.count[threadIdx.x]);
    __syncthreads();
    if (slot >= 0) base += t.base[slot];
    base = __shfl(base, leader < 0 ? lane : leader);
    if (valid) order[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
}

constexpr int GS_TOK = 64;    // tokens per workgroup (large N)
constexpr int GS_TOK_SMALL = 16;   // ... when N / 64 workgroups would leave most CUs idle
constexpr int GS_TOK_TINY = 4;     // ... one token per wave: a few thousand tokens still put two workgroups on every CU
constexpr int GS_TINY_N = 8192;    // below this many tokens the tiny form is used
constexpr int GS_SMALL_N = 32768;  // below this many tokens the small form is used (>= 2 workgroups per CU above)
__host__ __device__ inline int gs_tok(long long N) {
    return N < GS_TINY_N ? GS_TOK_TINY : (N < GS_SMALL_N ? GS_TOK_SMALL : GS_TOK);
}

// Wave 0 of a workgroup: group the workgroup's TOK tokens by code.  Group g = (s_gcode[g], member mask
// s_gmask[g]); groups are numbered by their first member.  counts (nullable) receives every group's size.
// (Cutting a hot code's group into sub-groups for the four waves was measured and rejected: the extra flushes
// contend on the same embed_sum row -- all tokens on one code: 113 -> 217 us at C3.)
template <int TOK>
__device__ __forceinline__ void build_groups(const int* s_code, unsigned long long* s_gmask, int* s_gcode,
                                             int* s_ngroups, int lane, float* counts, int stats_begin,
                                             int stats_count) {
    const int mine = lane < TOK ? s_code[lane] : -1;
    unsigned long long m = 0;
    for (int t = 0; t < TOK; ++t)
        if (s_code[t] == mine) m |= 1ull << t;
    const bool leader = mine >= 0 && (__ffsll(static_cast<long long>(m)) - 1) == lane;
    const unsigned long long lead_mask = __ballot(leader);
    if (leader) {
        const int slot = __popcll(lead_mask & ((1ull << lane) - 1ull));
        s_gmask[slot] = m;
        s_gcode[slot] = mine;
        const int sc = mine - stats_begin;
        if (counts && sc >= 0 && sc < stats_count) atomicAdd(counts + sc, static_cast<float>(__popcll(m)));
    }
    if (lane == 0) *s_ngroups = __popcll(lead_mask);
}

// key_sets > 1: the token's key is the minimum of keys[s * N + token], s < key_sets (the per-split slots the
// self-contained distance kernel writes).
template <int NCH, bool COSINE, int TOK>
__global__ __launch_bounds__(256) void gather_stats_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const long long* __restrict__ keys, int N, int K,
    int D, float* __restrict__ out_ste, float* __restrict__ out_q, float* __restrict__ out_diff,
    long long* __restrict__ idx_out, int stats_begin, int stats_count, float* __restrict__ counts,
    float* __restrict__ embed_sum, float* __restrict__ loss_partial, const int* __restrict__ perm, int key_sets) {
    extern __shared__ __attribute__((aligned(16))) float tbuf[];  // [4 waves][NCH*256] transposition rows
    __shared__ int s_code[TOK];
    __shared__ int s_tok[TOK];   // the workgroup's tokens: positions tok0.. of `perm` (tokens ordered by code), or tok0.. itself
    __shared__ unsigned long long s_gmask[TOK];
    __shared__ int s_gcode[TOK];
    __shared__ int s_ngroups;
    __shared__ float s_red[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tok0 = blockIdx.x * TOK;
    const int n_here = min(TOK, N - tok0);

    if (tid < TOK) {
        int code = -1, real = 0;
        if (tid < n_here) {
            real = perm ? perm[tok0 + tid] : tok0 + tid;
            long long key = keys[real];
            for (int ks = 1; ks < key_sets; ++ks) {
                const long long o = keys[static_cast<size_t>(ks) * N + real];
                key = o < key ? o : key;
            }
            code = key_index(key);
            code = code < 0 ? 0 : (code > K - 1 ? K - 1 : code);
            if (idx_out) idx_out[real] = code;
        }
        s_code[tid] = code;
        s_tok[tid] = real;
    }
    __syncthreads();
    if (wave == 0) build_groups<TOK>(s_code, s_gmask, s_gcode, &s_ngroups, lane, counts, stats_begin, stats_count);
    __syncthreads();
    const int n_groups = s_ngroups;

    const bool per_token = perm || TOK < GS_TOK;   // one loss term per token (else one per workgroup)
    float lsum = 0.f;  // MSE: sum (q-x)^2 over this lane's columns; COSINE: lane 0 holds sum of 1 - cos
    float* trow = tbuf + wave * (NCH * 256);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int g = wave; g < n_groups; g += 4) {
        const int code = __builtin_amdgcn_readfirstlane(s_gcode[g]);
        unsigned long long m = s_gmask[g];
        f32x4 q[NCH], sum[NCH];
        const f32x4* wrow = reinterpret_cast<const f32x4*>(w + static_cast<size_t>(code) * D);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c4 = j * 64 + lane;
            q[j] = (c4 * 4 < D) ? wrow[c4] : zero4;
            sum[j] = zero4;
        }
        float qq = 0.f;
        if (COSINE) {
#pragma unroll
            for (int j = 0; j < NCH; ++j)
                qq += q[j].x * q[j].x + q[j].y * q[j].y + q[j].z * q[j].z + q[j].w * q[j].w;
            qq = wave_sum(qq);
        }
        // the next member's row is requested before the current one is processed (a run of equal codes is a chain of
        // 3 KB rows from all over x: one row in flight per wave left the loads' latency in the open)
        f32x4 xn[NCH];
        int tn = __ffsll(static_cast<long long>(m)) - 1;
        m &= m - 1;
        {
            const f32x4* xr = reinterpret_cast<const f32x4*>(x + static_cast<size_t>(s_tok[tn]) * D);
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c4 = j * 64 + lane;
                xn[j] = (c4 * 4 < D) ? xr[c4] : zero4;
            }
        }
        while (tn >= 0) {
            const int t = tn;
            const size_t rowoff = static_cast<size_t>(s_tok[t]) * D;
            f32x4 xv[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) xv[j] = xn[j];
            tn = -1;
            if (m) {
                tn = __ffsll(static_cast<long long>(m)) - 1;
                m &= m - 1;
                const f32x4* xr = reinterpret_cast<const f32x4*>(x + static_cast<size_t>(s_tok[tn]) * D);
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int c4 = j * 64 + lane;
                    xn[j] = (c4 * 4 < D) ? xr[c4] : zero4;
                }
            }
            float pd = 0.f, px = 0.f;
            // ordered tokens: which tokens share a workgroup varies from run to run, so the loss is kept per TOKEN
            // and summed in token order afterwards -- the loss stays bit-reproducible.  Small batches (the 4- and
            // 16-token forms) keep it per token too: the term then does not depend on how tokens are grouped, and the
            // single-launch small-shape forward (t5vq_small.hip) produces the same terms
            float acc = per_token ? 0.f : lsum;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c4 = j * 64 + lane;
                const f32x4 df = q[j] - xv[j];
                if (c4 * 4 < D) {
                    if (out_ste) reinterpret_cast<f32x4*>(out_ste + rowoff)[c4] = xv[j] + df;
                    if (out_q) reinterpret_cast<f32x4*>(out_q + rowoff)[c4] = q[j];
                    if (out_diff) reinterpret_cast<f32x4*>(out_diff + rowoff)[c4] = df;
                }
                if (COSINE) {
                    pd += q[j].x * xv[j].x + q[j].y * xv[j].y + q[j].z * xv[j].z + q[j].w * xv[j].w;
                    px += xv[j].x * xv[j].x + xv[j].y * xv[j].y + xv[j].z * xv[j].z + xv[j].w * xv[j].w;
                } else {
                    acc = fmaf(df.x, df.x, acc);
                    acc = fmaf(df.y, df.y, acc);
                    acc = fmaf(df.z, df.z, acc);
                    acc = fmaf(df.w, df.w, acc);
                }
                sum[j] += xv[j];
            }
            if (COSINE) {
                pd = wave_sum(pd);
                px = wave_sum(px);
                const float den = sqrtf((qq + 1e-12f) * (px + 1e-12f));
                const float term = 1.0f - pd / den;
                if (per_token) {
                    if (lane == 0 && loss_partial) loss_partial[s_tok[t]] = term;
                } else if (lane == 0) {
                    lsum += term;
                }
            } else if (per_token) {
                acc = wave_sum(acc);
                if (lane == 0 && loss_partial) loss_partial[s_tok[t]] = acc;
            } else {
                lsum = acc;
            }
        }
        const int scode = code - stats_begin;
        if (embed_sum && scode >= 0 && scode < stats_count) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) reinterpret_cast<f32x4*>(trow)[j * 64 + lane] = sum[j];
            // same wave wrote and reads: LDS ops of one wave complete in order, no barrier needed
            float* dst = embed_sum + static_cast<size_t>(scode) * D;
#pragma unroll
            for (int mcol = 0; mcol < NCH * 4; ++mcol) {
                const int col = mcol * 64 + lane;
                if (col < D) atomicAdd(dst + col, trow[col]);
            }
        }
    }
    if (loss_partial && !per_token) {
        lsum = wave_sum(lsum);
        if (lane == 0) s_red[wave] = lsum;
        __syncthreads();
        if (tid == 0) loss_partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}

// Sum of n loss terms in a FIXED order (four strided fp64 accumulators per thread, wave butterflies, wave totals in
// wave order): the same value whichever kernel calls it.  1024 threads; the total is valid in thread 0.
__device__ __forceinline__ double sum_terms_1024(const float* __restrict__ p, long long n, double* s /*[16]*/) {
    // Thread t owns the groups of four terms #t, #t + 1024, ...; accumulator e takes element e of each group.  The
    // order depends on n alone: a 16-byte aligned pointer moves a group as one float4 (two in flight), any other
    // pointer reads the same elements one by one into the same accumulators.
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const long long n4 = n >> 2;
    if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
        const f32x4* q = reinterpret_cast<const f32x4*>(p);
        long long g = threadIdx.x;
        for (; g + 1024 < n4; g += 2048) {
            const f32x4 u = q[g], v = q[g + 1024];
            a0 += static_cast<double>(u.x); a1 += static_cast<double>(u.y);
            a2 += static_cast<double>(u.z); a3 += static_cast<double>(u.w);
            a0 += static_cast<double>(v.x); a1 += static_cast<double>(v.y);
            a2 += static_cast<double>(v.z); a3 += static_cast<double>(v.w);
        }
        if (g < n4) {
            const f32x4 u = q[g];
            a0 += static_cast<double>(u.x); a1 += static_cast<double>(u.y);
            a2 += static_cast<double>(u.z); a3 += static_cast<double>(u.w);
        }
    } else {
        for (long long g = threadIdx.x; g < n4; g += 1024) {
            a0 += static_cast<double>(p[4 * g]); a1 += static_cast<double>(p[4 * g + 1]);
            a2 += static_cast<double>(p[4 * g + 2]); a3 += static_cast<double>(p[4 * g + 3]);
        }
    }
    const long long tail = n4 << 2;   // the last n % 4 terms: threads 0..2, accumulator 0
    if (tail + threadIdx.x < n) a0 += static_cast<double>(p[tail + threadIdx.x]);
    const double acc = wave_sum_d((a0 + a1) + (a2 + a3));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int k = 0; k < 16; ++k) t += s[k];
    return t;
}

__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ p, long long n,
                                                              float* __restrict__ out) {
    __shared__ double s[16];
    const double t = sum_terms_1024(p, n, s);
    if (threadIdx.x == 0) out[0] = static_cast<float>(t);
}

// ------------------------------------------------------------------------------------------------
// K4a: cluster-size EMA + Laplace smoothing + loss / perplexity scalars (one workgroup)
// ------------------------------------------------------------------------------------------------
__device__ double block_sum_d(double v, double* s) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < static_cast<int>(blockDim.x >> 6); ++i) t += s[i];
    return t;
}

// mode 0: scalars only (eval).  mode 1: fused update.  mode 2: phase 1 of a code-sharded update
// (cs <- cs*decay + omd*counts; scalars[1] = local sum p log(p+1e-10), scalars[2] = local sum cs).
// mode 3: phase 2 (reads the all-reduced scalars[1], scalars[2]; normalises cs; writes perplexity).
// loss_terms[0 .. n_terms) are summed here (n_terms == 1: a ready sum) -- the forward leaves its loss terms
// unreduced, which saves the reduce launch.  VEC: cs and counts are 16-byte aligned and K % 4 == 0, so each thread
// walks float4s (K = 10000: 3 dependent round trips instead of 10).
template <bool VEC>
__global__ __launch_bounds__(1024) void ema_cluster_kernel(float* __restrict__ cs, const float* __restrict__ counts,
                                                           const float* __restrict__ loss_terms, long long n_terms,
                                                           int K, float n_tokens, float loss_denom, float decay,
                                                           float omd, float eps, float k_eps, float cc,
                                                           int plain_variant, int mode, float* __restrict__ scalars) {
    __shared__ double s[16];
    double ent = 0.0, nsum = 0.0;
    float n = 0.f;
    const bool upd = mode != 0 && cs;
    auto one = [&](float c, float old) {
        const float p = __fdiv_rn(c, n_tokens);
        ent += static_cast<double>(__fmul_rn(p, logf(__fadd_rn(p, 1e-10f))));
        const float v = __fadd_rn(__fmul_rn(old, decay), __fmul_rn(omd, c));
        if (upd) nsum += static_cast<double>(v);
        return v;
    };
    if (mode != 3) {
        if (VEC) {
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            for (int k4 = threadIdx.x; k4 < (K >> 2); k4 += blockDim.x) {
                const f32x4 c = reinterpret_cast<const f32x4*>(counts)[k4];
                const f32x4 o = upd ? reinterpret_cast<const f32x4*>(cs)[k4] : zero4;
                f32x4 v;
                v.x = one(c.x, o.x), v.y = one(c.y, o.y), v.z = one(c.z, o.z), v.w = one(c.w, o.w);
                if (upd) reinterpret_cast<f32x4*>(cs)[k4] = v;
            }
        } else {
            for (int k = threadIdx.x; k < K; k += blockDim.x) {
                const float v = one(counts[k], upd ? cs[k] : 0.f);
                if (upd) cs[k] = v;
            }
        }
        ent = block_sum_d(ent, s);
        if (mode != 0) n = static_cast<float>(block_sum_d(nsum, s));
    } else {
        ent = static_cast<double>(scalars[1]);
        n = scalars[2];
        __syncthreads();  // everyone has read the scalars before thread 0 overwrites them
    }
    if ((mode == 1 || mode == 3) && cs) {
        const float den = __fadd_rn(n, k_eps);
        if (VEC) {
            for (int k4 = threadIdx.x; k4 < (K >> 2); k4 += blockDim.x) {  // each thread re-reads only what it wrote
                f32x4 v = reinterpret_cast<f32x4*>(cs)[k4];
                v.x = __fmul_rn(__fdiv_rn(__fadd_rn(v.x, eps), den), n);
                v.y = __fmul_rn(__fdiv_rn(__fadd_rn(v.y, eps), den), n);
                v.z = __fmul_rn(__fdiv_rn(__fadd_rn(v.z, eps), den), n);
                v.w = __fmul_rn(__fdiv_rn(__fadd_rn(v.w, eps), den), n);
                reinterpret_cast<f32x4*>(cs)[k4] = v;
            }
        } else {
            for (int k = threadIdx.x; k < K; k += blockDim.x)  // mode 1: each thread re-reads only what it wrote
                cs[k] = __fmul_rn(__fdiv_rn(__fadd_rn(cs[k], eps), den), n);
        }
    }
    double lsum = 0.0;
    if (loss_terms) lsum = n_terms == 1 ? static_cast<double>(loss_terms[0]) : sum_terms_1024(loss_terms, n_terms, s);
    if (threadIdx.x == 0) {
        const float m = loss_terms ? __fdiv_rn(static_cast<float>(lsum), loss_denom) : 0.f;
        const float cm = __fmul_rn(cc, m);
        scalars[0] = plain_variant ? __fadd_rn(m, cm) : cm;
        scalars[1] = mode == 2 ? static_cast<float>(ent) : expf(static_cast<float>(-ent));
        scalars[2] = n;
        scalars[3] = m;
    }
}

// The fp16 image of the code row a wave has just written (t5vq_internal.h; the filtered search of the NEXT call then
// reads no fp32 codebook row for its preparation): what filter_prep_kernel would make of the row, bit for bit -- the
// same scale rule on the same values, sqrt of the same |e|^2.  qv: the row in registers, float4 #(lane + 64 j)
// (D <= 2048: the launcher refuses an image beyond); m: the lane's largest magnitude, sq: |e_row|^2.
template <int NCH>
__device__ __forceinline__ void emit_code_image_row(const T5vqCodeImage& img, const f32x4 (&qv)[NCH], long long row,
                                                    int D, int lane, float m, float sq) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float invv;
    const float scale = t5vq_image_scale(m, &invv);
    if (lane == 0) {
        img.inv[row] = invv;
        img.se[row] = sqrtf(sq);
    }
    _Float16* dst = img.rows + row * D;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int i = lane + 64 * j;
        if (i < (D >> 2)) t5vq_image_store4(dst, i, qv[j].x, qv[j].y, qv[j].z, qv[j].w, scale);
    }
}

// K4b: codebook EMA, one wave per code row
// ee_out (nullable): |e_row|^2 of the row just written, accumulated in row_sqnorm_wave's lane layout and order from
// the values in registers -- bit-identical to t5vq_row_sqnorms_f32 run on the new codebook, so the next call need not
// read the codebook for its norms (vqvae/vqvae.py:107 recomputes them every call).
template <int NCH>   // NCH > 0: also write the row's fp16 image, the row held in NCH float4 per lane (D <= 256 NCH); 0: no image
__global__ __launch_bounds__(256) void ema_embed_kernel(float* __restrict__ ema_w, float* __restrict__ w,
                                                        const float* __restrict__ dw, const float* __restrict__ cs,
                                                        int K, int D, float decay, float omd,
                                                        float* __restrict__ ee_out, T5vqCodeImage img, int kpad) {
    const int lane = threadIdx.x & 63;
    const long long row = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (NCH > 0 && blockIdx.x == 0)   // the image's padding entries: inverse scale 1, sqrt(|e|^2) = +inf
        for (int r = K + threadIdx.x; r < kpad; r += blockDim.x) {
            img.inv[r] = 1.0f;
            img.se[r] = __int_as_float(0x7F800000);
        }
    if (row >= K) return;
    const float c = cs[row];
    f32x4* e = reinterpret_cast<f32x4*>(ema_w + row * D);
    f32x4* o = reinterpret_cast<f32x4*>(w + row * D);
    const f32x4* s = reinterpret_cast<const f32x4*>(dw + row * D);
    float sq = 0.f, m = 0.f;
    auto one = [&](int i) {
        const f32x4 a = e[i], b = s[i];
        f32x4 v, q;
        v.x = __fadd_rn(__fmul_rn(a.x, decay), __fmul_rn(omd, b.x));
        v.y = __fadd_rn(__fmul_rn(a.y, decay), __fmul_rn(omd, b.y));
        v.z = __fadd_rn(__fmul_rn(a.z, decay), __fmul_rn(omd, b.z));
        v.w = __fadd_rn(__fmul_rn(a.w, decay), __fmul_rn(omd, b.w));
        q.x = __fdiv_rn(v.x, c);
        q.y = __fdiv_rn(v.y, c);
        q.z = __fdiv_rn(v.z, c);
        q.w = __fdiv_rn(v.w, c);
        e[i] = v;
        o[i] = q;
        sq = sq4(q, sq);
        return q;
    };
    if constexpr (NCH > 0) {   // the row stays in registers for its image
        f32x4 qv[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int i = lane + 64 * j;
            if (i < (D >> 2)) {
                qv[j] = one(i);
                m = t5vq_absmax4(m, qv[j].x, qv[j].y, qv[j].z, qv[j].w);
            }
        }
        sq = wave_sum(sq);
        if (ee_out && lane == 0) ee_out[row] = sq;
        emit_code_image_row<NCH>(img, qv, row, D, lane, m, sq);
        return;
    }
    for (int i = lane; i < (D >> 2); i += 64) one(i);
    if (ee_out) {
        sq = wave_sum(sq);
        if (lane == 0) ee_out[row] = sq;
    }
}

// ------------------------------------------------------------------------------------------------
// Encoder tail fused into the VQ prologue: T5LayerNorm (RMS norm, no mean subtraction, learned
// scale) + dropout produce the VQ input; the same pass also emits the row norms |x|^2 the distance kernel needs.
// One wave per row; lane layout as everywhere: float4 #(j*64 + lane).
// ------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_sqnorms_kernel(const float* __restrict__ hid,
                                                              const float* __restrict__ gamma,
                                                              const unsigned char* __restrict__ keep, float scale,
                                                              float eps, int N, int D, float* __restrict__ x_out,
                                                              float* __restrict__ xx, float* __restrict__ rms_out,
                                                              T5vqTokenImage img) {
    const int lane = threadIdx.x & 63;
    const long long row = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const f32x4* hr = reinterpret_cast<const f32x4*>(hid + row * D);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 hv[NCH];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c4 = j * 64 + lane;
        hv[j] = (c4 * 4 < D) ? hr[c4] : zero4;
        ss = fmaf(hv[j].x, hv[j].x, ss);
        ss = fmaf(hv[j].y, hv[j].y, ss);
        ss = fmaf(hv[j].z, hv[j].z, ss);
        ss = fmaf(hv[j].w, hv[j].w, ss);
    }
    ss = wave_sum(ss);
    const float var = __fdiv_rn(ss, static_cast<float>(D));   // x.pow(2).mean(-1)
    const float r = __fsqrt_rn(__fadd_rn(var, eps));           // sqrt(variance + eps)
    if (rms_out && lane == 0) rms_out[row] = r;
    float s2 = 0.f, m = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c4 = j * 64 + lane;
        if (c4 * 4 < D) {
            const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[c4];
            f32x4& y = hv[j];   // the input row is done with: its registers hold the output row from here on
            y.x = __fmul_rn(g.x, __fdiv_rn(y.x, r));           // weight * (x / sqrt(...))
            y.y = __fmul_rn(g.y, __fdiv_rn(y.y, r));
            y.z = __fmul_rn(g.z, __fdiv_rn(y.z, r));
            y.w = __fmul_rn(g.w, __fdiv_rn(y.w, r));
            if (keep) {
                const uchar4 m = reinterpret_cast<const uchar4*>(keep + row * D)[c4];
                y.x = m.x ? __fmul_rn(y.x, scale) : 0.f;
                y.y = m.y ? __fmul_rn(y.y, scale) : 0.f;
                y.z = m.z ? __fmul_rn(y.z, scale) : 0.f;
                y.w = m.w ? __fmul_rn(y.w, scale) : 0.f;
            }
            reinterpret_cast<f32x4*>(x_out + row * D)[c4] = y;
            s2 = fmaf(y.x, y.x, s2);
            s2 = fmaf(y.y, y.y, s2);
            s2 = fmaf(y.z, y.z, s2);
            s2 = fmaf(y.w, y.w, s2);
            m = t5vq_absmax4(m, y.x, y.y, y.z, y.w);
        }
    }
    s2 = wave_sum(s2);
    if (xx && lane == 0) xx[row] = s2;
    if (img.rows) {   // the row's fp16 image for the filtered search, as filter_prep_kernel would make it
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float invv;
        const float sc = t5vq_image_scale(m, &invv);
        if (lane == 0) img.inv[row] = invv;
        _Float16* dst = img.rows + row * D;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c4 = j * 64 + lane;
            if (c4 * 4 < D) t5vq_image_store4(dst, c4, hv[j].x, hv[j].y, hv[j].z, hv[j].w, sc);
        }
    }
}

// Backward of the tail.  With gy = g * keep * scale, a = gy * gamma, r = sqrt(mean(h^2) + eps):
//   grad_h = a / r - h * (sum_d a_d h_d) / (D r^3);   grad_gamma = sum_rows gy * h / r.
// A workgroup walks rows [blockIdx, blockIdx + gridDim, ...) (4 at a time, one per wave) and leaves ONE partial
// grad_gamma row in gw_partial[blockIdx]; rmsnorm_gamma_reduce_kernel sums the partials in fixed order.
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_backward_kernel(const float* __restrict__ g,
                                                               const float* __restrict__ hid,
                                                               const float* __restrict__ gamma,
                                                               const unsigned char* __restrict__ keep, float scale,
                                                               const float* __restrict__ rms, int N, int D,
                                                               float* __restrict__ grad_h,
                                                               float* __restrict__ gw_partial) {
    extern __shared__ __attribute__((aligned(16))) float tbuf[];  // [4 waves][NCH*256]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 gam[NCH], gw[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c4 = j * 64 + lane;
        gam[j] = (c4 * 4 < D) ? reinterpret_cast<const f32x4*>(gamma)[c4] : zero4;
        gw[j] = zero4;
    }
    const float inv_d = 1.0f / static_cast<float>(D);
    for (long long row = static_cast<long long>(blockIdx.x) * 4 + wave; row < N;
         row += static_cast<long long>(gridDim.x) * 4) {
        const float r = rms[row];
        const float inv_r = 1.0f / r;
        f32x4 a[NCH], hv[NCH];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c4 = j * 64 + lane;
            a[j] = zero4;
            hv[j] = zero4;
            if (c4 * 4 < D) {
                f32x4 gy = reinterpret_cast<const f32x4*>(g + row * D)[c4];
                hv[j] = reinterpret_cast<const f32x4*>(hid + row * D)[c4];
                if (keep) {
                    const uchar4 m = reinterpret_cast<const uchar4*>(keep + row * D)[c4];
                    gy.x = m.x ? gy.x * scale : 0.f;
                    gy.y = m.y ? gy.y * scale : 0.f;
                    gy.z = m.z ? gy.z * scale : 0.f;
                    gy.w = m.w ? gy.w * scale : 0.f;
                }
                a[j] = gy * gam[j];
                dot += a[j].x * hv[j].x + a[j].y * hv[j].y + a[j].z * hv[j].z + a[j].w * hv[j].w;
                gw[j] += gy * hv[j] * inv_r;
            }
        }
        dot = wave_sum(dot);
        const float c = dot * inv_d * inv_r * inv_r * inv_r;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c4 = j * 64 + lane;
            if (c4 * 4 < D) reinterpret_cast<f32x4*>(grad_h + row * D)[c4] = a[j] * inv_r - hv[j] * c;
        }
    }
    if (!gw_partial) return;
    // sum the four waves' partial grad_gamma rows through LDS, fixed order
#pragma unroll
    for (int j = 0; j < NCH; ++j) reinterpret_cast<f32x4*>(tbuf + wave * (NCH * 256))[j * 64 + lane] = gw[j];
    __syncthreads();
    for (int col = threadIdx.x; col < D; col += 256) {
        const float v = (tbuf[col] + tbuf[NCH * 256 + col]) + (tbuf[2 * NCH * 256 + col] + tbuf[3 * NCH * 256 + col]);
        gw_partial[static_cast<size_t>(blockIdx.x) * D + col] = v;
    }
}

__global__ __launch_bounds__(256) void rmsnorm_gamma_reduce_kernel(const float* __restrict__ gw_partial, int G, int D,
                                                                   float* __restrict__ grad_gamma) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= D) return;
    float acc = 0.f;
    for (int b = 0; b < G; ++b) acc += gw_partial[static_cast<size_t>(b) * D + col];
    grad_gamma[col] = acc;
}

// ------------------------------------------------------------------------------------------------
// Order-independent EMA statistics (opt-in): embed_sum accumulated in 64-bit fixed point.  Integer addition is
// associative, so the result does not depend on the order in which workgroups (or GPUs: the all-reduce is an
// integer SUM) arrive -- bit-reproducible from run to run and for any number of ranks; and it is the EXACT sum of
// the quantised inputs, rounded once when converted back.  Fixed point: x * 2^s with 2^s * max|x| <= 2^38 (38
// fractional bits relative to the largest input magnitude; up to 2^24 tokens per code fit 63 bits).  max|x| is
// bounded by sqrt(max_i |x_i|^2), taken from the row norms.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void max_norm_kernel(const float* __restrict__ xx, long long n,
                                                       int* __restrict__ max_bits) {   // *max_bits zeroed before
    float m = 0.f;
    for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < n;
         i += static_cast<long long>(gridDim.x) * 256)
        m = fmaxf(m, xx[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(max_bits, __float_as_int(m));   // xx >= 0: int order = float order
}

__device__ __forceinline__ int fixed_shift(int max_xx_bits) {   // s with 2^s * sqrt(max xx) <= 2^38
    const int e2 = ((max_xx_bits >> 23) & 0xFF) - 127;            // floor(log2(max xx)); max|x| < 2^ceil((e2+1)/2)
    int s = 38 - ((e2 + 2) >> 1);
    return s > 1000 ? 1000 : (s < -1000 ? -1000 : s);
}

template <int NCH, int TOK>
__global__ __launch_bounds__(256) void embed_sum_fixed_kernel(const float* __restrict__ x,
                                                              const long long* __restrict__ keys, int N, int K, int D,
                                                              const int* __restrict__ max_xx_bits, int stats_begin,
                                                              int stats_count, long long* __restrict__ embed_sum) {
    extern __shared__ __attribute__((aligned(16))) long long lbuf[];  // [4 waves][NCH*256] transposition rows
    __shared__ int s_code[TOK];
    __shared__ unsigned long long s_gmask[TOK];
    __shared__ int s_gcode[TOK];
    __shared__ int s_ngroups;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok0 = blockIdx.x * TOK;
    const int n_here = min(TOK, N - tok0);
    if (tid < TOK) {
        int code = -1;
        if (tid < n_here) {
            code = key_index(keys[tok0 + tid]);
            code = code < 0 ? 0 : (code > K - 1 ? K - 1 : code);
        }
        s_code[tid] = code;
    }
    __syncthreads();
    if (wave == 0) build_groups<TOK>(s_code, s_gmask, s_gcode, &s_ngroups, lane, nullptr, 0, 0);
    __syncthreads();
    const int n_groups = s_ngroups;
    const double scale = ldexp(1.0, fixed_shift(*max_xx_bits));
    long long* trow = lbuf + wave * (NCH * 256);
    for (int g = wave; g < n_groups; g += 4) {
        const int scode = __builtin_amdgcn_readfirstlane(s_gcode[g]) - stats_begin;
        if (scode < 0 || scode >= stats_count) continue;
        unsigned long long m = s_gmask[g];
        long long sum[NCH * 4];
#pragma unroll
        for (int j = 0; j < NCH * 4; ++j) sum[j] = 0;
        while (m) {
            const int t = __ffsll(static_cast<long long>(m)) - 1;
            m &= m - 1;
            const f32x4* xr = reinterpret_cast<const f32x4*>(x + static_cast<size_t>(tok0 + t) * D);
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c4 = j * 64 + lane;
                if (c4 * 4 < D) {
                    const f32x4 v = xr[c4];
                    sum[4 * j + 0] += __double2ll_rn(static_cast<double>(v.x) * scale);
                    sum[4 * j + 1] += __double2ll_rn(static_cast<double>(v.y) * scale);
                    sum[4 * j + 2] += __double2ll_rn(static_cast<double>(v.z) * scale);
                    sum[4 * j + 3] += __double2ll_rn(static_cast<double>(v.w) * scale);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) trow[(j * 64 + lane) * 4 + c] = sum[4 * j + c];
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(embed_sum) + static_cast<size_t>(scode) * D;
#pragma unroll
        for (int mcol = 0; mcol < NCH * 4; ++mcol) {
            const int col = mcol * 64 + lane;
            if (col < D) atomicAdd(dst + col, static_cast<unsigned long long>(trow[col]));
        }
    }
}

// K4b for the fixed-point statistics: dw = the exact sum, rounded once
template <int NCH>
__global__ __launch_bounds__(256) void ema_embed_fixed_kernel(float* __restrict__ ema_w, float* __restrict__ w,
                                                              const long long* __restrict__ dw,
                                                              const int* __restrict__ max_xx_bits,
                                                              const float* __restrict__ cs, int K, int D, float decay,
                                                              float omd, float* __restrict__ ee_out, T5vqCodeImage img,
                                                              int kpad) {
    const int lane = threadIdx.x & 63;
    const long long row = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (NCH > 0 && blockIdx.x == 0)
        for (int r = K + threadIdx.x; r < kpad; r += blockDim.x) {
            img.inv[r] = 1.0f;
            img.se[r] = __int_as_float(0x7F800000);
        }
    if (row >= K) return;
    const double inv = ldexp(1.0, -fixed_shift(*max_xx_bits));
    const float c = cs[row];
    float sq = 0.f, m = 0.f;
    auto one = [&](int i4) {   // float4 #i4 per lane: row_sqnorm_wave's layout
        f32x4 q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long i = row * D + 4 * i4 + e;
            const float b = static_cast<float>(static_cast<double>(dw[i]) * inv);
            const float v = __fadd_rn(__fmul_rn(ema_w[i], decay), __fmul_rn(omd, b));
            ema_w[i] = v;
            q[e] = __fdiv_rn(v, c);
            w[i] = q[e];
        }
        sq = sq4(q, sq);
        return q;
    };
    if constexpr (NCH > 0) {   // the row stays in registers for its image
        f32x4 qv[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int i4 = lane + 64 * j;
            if (i4 < (D >> 2)) {
                qv[j] = one(i4);
                m = t5vq_absmax4(m, qv[j].x, qv[j].y, qv[j].z, qv[j].w);
            }
        }
        sq = wave_sum(sq);
        if (ee_out && lane == 0) ee_out[row] = sq;
        emit_code_image_row<NCH>(img, qv, row, D, lane, m, sq);
        return;
    }
    for (int i4 = lane; i4 < (D >> 2); i4 += 64) one(i4);
    if (ee_out) {
        sq = wave_sum(sq);
        if (lane == 0) ee_out[row] = sq;
    }
}

// ------------------------------------------------------------------------------------------------
// backward helpers
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ste_backward_kernel(const float* __restrict__ g_out,
                                                           const float* __restrict__ diff,
                                                           const float* __restrict__ g_loss, float coef, long long n4,
                                                           float* __restrict__ grad_x) {
    const float s = g_loss ? g_loss[0] * coef : 0.f;
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 d = reinterpret_cast<const f32x4*>(diff)[i];
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (g_out) g = reinterpret_cast<const f32x4*>(g_out)[i];
        g.x = fmaf(-s, d.x, g.x);
        g.y = fmaf(-s, d.y, g.y);
        g.z = fmaf(-s, d.z, g.z);
        g.w = fmaf(-s, d.w, g.w);
        reinterpret_cast<f32x4*>(grad_x)[i] = g;
    }
}

// scatter-add of rows by index with the same in-LDS grouping as gather_stats
template <int NCH>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ rows,
                                                           const long long* __restrict__ idx,
                                                           const float* __restrict__ scale_dev, float scale_host,
                                                           int N, int K, int D, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float tbuf[];
    __shared__ int s_code[GS_TOK];
    __shared__ unsigned long long s_gmask[GS_TOK];
    __shared__ int s_gcode[GS_TOK];
    __shared__ int s_ngroups;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok0 = blockIdx.x * GS_TOK;
    const int n_here = min(GS_TOK, N - tok0);
    const float scale = scale_host * (scale_dev ? scale_dev[0] : 1.0f);
    if (tid < GS_TOK) {
        int code = -1;
        if (tid < n_here) {
            long long c = idx[tok0 + tid];
            code = c < 0 ? 0 : (c > K - 1 ? K - 1 : static_cast<int>(c));
        }
        s_code[tid] = code;
    }
    __syncthreads();
    if (wave == 0) build_groups<GS_TOK>(s_code, s_gmask, s_gcode, &s_ngroups, lane, nullptr, 0, 0);
    __syncthreads();
    const int n_groups = s_ngroups;
    float* trow = tbuf + wave * (NCH * 256);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int g = wave; g < n_groups; g += 4) {
        const int code = __builtin_amdgcn_readfirstlane(s_gcode[g]);
        unsigned long long m = s_gmask[g];
        f32x4 sum[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) sum[j] = zero4;
        while (m) {
            const int t = __ffsll(static_cast<long long>(m)) - 1;
            m &= m - 1;
            const f32x4* r = reinterpret_cast<const f32x4*>(rows + static_cast<size_t>(tok0 + t) * D);
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c4 = j * 64 + lane;
                if (c4 * 4 < D) sum[j] += r[c4];
            }
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) reinterpret_cast<f32x4*>(trow)[j * 64 + lane] = sum[j] * scale;
        float* dst = out + static_cast<size_t>(code) * D;
#pragma unroll
        for (int mcol = 0; mcol < NCH * 4; ++mcol) {
            const int col = mcol * 64 + lane;
            if (col < D) atomicAdd(dst + col, trow[col]);
        }
    }
}

// cosine latent-loss gradients, one wave per token row
__global__ __launch_bounds__(256) void cosine_backward_kernel(const float* __restrict__ x, const float* __restrict__ q,
                                                              const float* __restrict__ g_out,
                                                              const float* __restrict__ g_loss, float coef_x,
                                                              float coef_q, int N, int D, float* __restrict__ grad_x,
                                                              float* __restrict__ grad_q) {
    const int lane = threadIdx.x & 63;
    const long long row = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float gl = g_loss ? g_loss[0] : 0.f;
    const float sx = gl * coef_x, sq = gl * coef_q;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * D);
    const f32x4* qr = reinterpret_cast<const f32x4*>(q + row * D);
    float pd = 0.f, px = 0.f, pq = 0.f;
    for (int i = lane; i < (D >> 2); i += 64) {
        const f32x4 a = qr[i], b = xr[i];
        pd += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        pq += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        px += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
    }
    pd = wave_sum(pd);
    const float m1 = wave_sum(pq) + 1e-12f;  // |q|^2 + eps
    const float m2 = wave_sum(px) + 1e-12f;  // |x|^2 + eps
    const float den = sqrtf(m1 * m2);
    // c = 1 - pd/den;  dc/dx = -q/den + pd * x / (den * m2);  dc/dq = -x/den + pd * q / (den * m1)
    const float inv = 1.0f / den;
    const float kx = pd * inv / m2, kq = pd * inv / m1;
    for (int i = lane; i < (D >> 2); i += 64) {
        const f32x4 a = qr[i], b = xr[i];
        f32x4 gx = (b * kx - a * inv) * sx;
        if (g_out) gx += reinterpret_cast<const f32x4*>(g_out + row * D)[i];
        reinterpret_cast<f32x4*>(grad_x + row * D)[i] = gx;
        if (grad_q) reinterpret_cast<f32x4*>(grad_q + row * D)[i] = (a * kq - b * inv) * sq;
    }
}

__global__ void one_hot_kernel(const long long* __restrict__ idx, long long N, long long K, float* __restrict__ enc) {
    long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= N) return;
    long long c = idx[i];
    if (c >= 0 && c < K) enc[i * K + c] = 1.0f;
}

int pick_nch(int64_t D) {
    const int64_t need = (D + 255) / 256;
    if (need <= 1) return 1;
    if (need <= 2) return 2;
    if (need <= 3) return 3;
    if (need <= 4) return 4;
    if (need <= 8) return 8;
    return 0;
}

// Tuning knobs (t5vq_set_tuning).  Per thread: a diagnostic call on one thread never changes what another
// thread's launches pick, and nothing here is read and written concurrently.
thread_local int g_dist_variant = 0;  // 0 = default configuration
thread_local int g_force_splits = 0;  // 0 = heuristic
thread_local int g_depth2 = 1;        // staging registers double-buffered (loads two chunks ahead)

using CfgDefault = DistCfg<4, 32, false, 2>;
using CfgSmall = DistCfg<2, 32, false, 2>;   // 128 x 64 tile: twice the workgroups when the problem cannot fill the chip
#ifdef T5VQ_TUNING
using Cfg1 = DistCfg<4, 32, true, 2>;
using Cfg2 = DistCfg<8, 16, false, 2>;
using Cfg3 = DistCfg<8, 16, true, 2>;
using Cfg4 = DistCfg<4, 16, true, 3>;
using Cfg5 = DistCfg<8, 32, true, 1>;
using Cfg6 = DistCfg<4, 16, false, 3>;
using Cfg7 = DistCfg<8, 32, false, 1>;
using Cfg8 = DistCfg<4, 32, false, 2, 1>;
using Cfg9 = DistCfg<4, 32, false, 2, 2>;
using Cfg10 = DistCfg<4, 32, false, 2, 3>;
using Cfg11 = DistCfg<8, 32, false, 1, 1>;
using Cfg12 = DistCfg<8, 32, false, 1, 2>;
using Cfg13 = DistCfg<8, 32, false, 1, 3>;
using Cfg14 = DistCfg<4, 32, false, 2, 4>;
using Cfg15 = DistCfg<4, 32, false, 2, 5>;
#endif

// Code splits.  The S workgroups of one token tile that cover different code ranges are given block ids 8 apart,
// i.e. ONE XCD, and share the X tile in its 4 MiB L2 (at the price of one more pass over W per round of
// resident workgroups).  Two criteria, in this order:
//  1. time: each of the 256 CUs has one matrix pipe per SIMD that its (up to two) resident workgroups share, so
//     a launch takes about ceil(WGs / 256) x (t + 0.1) tile-times, t = code tiles per split (0.1: prologue and
//     epilogue of a workgroup) -- fills the chip at small N, avoids a nearly empty last round;
//  2. among choices within 2 % of the best time, the fewest modelled L2-miss bytes (64 resident workgroups per
//     XCD, ~2.5 MB of L2 left for X tiles).  Measured at C3 (rocprofv3 FETCH_SIZE x 2): 12.9 GB unsplit ->
//     4.7 GB at 8 splits, kernel time -1 %.
// Returns the number of splits; *time_out = modelled time in units of one 128-code tile on one pipe.
// ranged: the splits divide each token tile's OWN admissible code range (dist_mfma_v2_kernel), so more splits mean
// finer, better balanced pieces whatever the role layout -- among the choices within 2 % of the best modelled time
// the LARGEST number of splits is taken (8 roles of 1024 codes at N = 65536: 0.96 -> 0.81 ms).
int64_t plan_splits(int64_t N, int64_t K, int64_t D, int code_tile, double* time_out, bool ranged = false) {
    const int64_t tok_tiles = (N + TOK_TILE - 1) / TOK_TILE;
    const int64_t code_tiles = (K + code_tile - 1) / code_tile;
    const double tile_cost = code_tile / 128.0;
    int64_t splits = 1;
    double best_time = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
        double best_bytes = 0.0;
        bool have = false;
        for (int64_t t = 1; t <= code_tiles; ++t) {
            const int64_t s = (code_tiles + t - 1) / t;
            if (s > 256 || (t > 1 && (code_tiles + t - 2) / (t - 1) == s)) continue;  // same s with fewer tiles
            const double rounds = static_cast<double>((tok_tiles * s + 255) / 256);
            const double time = rounds * (static_cast<double>(t) + 0.1) * tile_cost;
            if (pass == 0) {
                if (best_time == 0.0 || time < best_time) best_time = time;
                continue;
            }
            if (time > best_time * 1.02) continue;
            if (ranged) {
                if (s > splits) splits = s;
                continue;
            }
            const double tiles_per_xcd = s >= 64 ? 1.0 : 64.0 / static_cast<double>(s);
            const double x_ws = tiles_per_xcd * TOK_TILE * static_cast<double>(D) * 4.0;
            const double miss = x_ws > 2.5e6 ? 1.0 - 2.5e6 / x_ws : 0.0;
            const double x_bytes = static_cast<double>(N) * D * 4.0 * (1.0 + (static_cast<double>(t) - 1.0) * miss);
            double w_rounds = static_cast<double>(tok_tiles) * s / 64.0;
            if (w_rounds < 1.0) w_rounds = 1.0;
            const double bytes = x_bytes + w_rounds * static_cast<double>(K) * D * 4.0;
            if (!have || bytes < best_bytes) {
                have = true;
                best_bytes = bytes;
                splits = s;
            }
        }
    }
    if (time_out) *time_out = best_time;
    return splits;
}

// Variations of a distance launch that the plain entry points do not need
struct DistOpts {
    bool keys_prefilled = false;     // composite forward: keys already hold KEY_NONE (no fill launch before a split)
    bool bounded = false;            // top-k pass: the range argument holds one excluded key per token
    const int* tok_list = nullptr;   // decision scan of the filtered search: device list of the tokens to search ...
    const int* n_listed = nullptr;   // ... and its length (device)
    const int* order = nullptr;      // ranged search: the tokens in role order (t5vq_order_by_range); speed only
};

template <class C, bool V2 = false>
int launch_dist_cfg(const float* x, const float* w, const float* xx, const float* ee, const int64_t* range,
                           int64_t N, int64_t K, int64_t D, int64_t code_offset, int merge, int64_t* keys,
                           float* dist, hipStream_t stream, const DistOpts& opts = DistOpts()) {
    // per (tile configuration, device) once; two host threads that race here both set the same attributes (idempotent)
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    T5VQ_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_done[dev].load(std::memory_order_acquire)) {
        T5VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_mfma_kernel<C, false, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        T5VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_mfma_kernel<C, true, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        T5VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_mfma_kernel<C, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        T5VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_mfma_kernel<C, false, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        T5VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_mfma_kernel<C, false, false, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        T5VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_mfma_kernel<C, true, false, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        if (V2) {
            T5VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_mfma_v2_kernel<C, false, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
            T5VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_mfma_v2_kernel<C, true, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
            T5VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_mfma_v2_kernel<C, false, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
            T5VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_mfma_v2_kernel<C, true, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        }
        attr_done[dev].store(true, std::memory_order_release);
    }
    const int64_t tok_tiles = (N + TOK_TILE - 1) / TOK_TILE;
    const int64_t code_tiles = (K + C::CODE_TILE - 1) / C::CODE_TILE;
    int64_t splits = 1;
    // the restructured main loop (and with it the ranged form that divides each token tile's OWN code range among the
    // splits) runs only for these shapes; the general kernel cuts the whole codebook statically
    const bool v2 = V2 && !dist && !opts.bounded && !opts.tok_list && D % C::BKC == 0;
    if (!dist) {
        splits = plan_splits(N, K, D, C::CODE_TILE, nullptr, range != nullptr && v2);
        if (g_force_splits > 0) {
            splits = g_force_splits;
            while (splits > code_tiles) splits /= 2;
        }
        if (opts.tok_list) splits = code_tiles;   // the list is usually short: spread its code tiles over the chip
    }
    const int64_t tiles_per_split = (code_tiles + splits - 1) / splits;
    const int atomic = (splits > 1 || merge) ? 1 : 0;
    if (!dist && splits > 1 && !merge && !opts.keys_prefilled && !opts.tok_list) {
        hipLaunchKernelGGL(fill_keys_kernel, dim3(static_cast<unsigned>((N + 255) / 256)), dim3(256), 0, stream,
                           reinterpret_cast<long long*>(keys), static_cast<long long>(N));
        T5VQ_CHECK_HIP(hipGetLastError());
    }
    int64_t n_blocks = (tok_tiles + 7) / 8 * 8 * splits;
    if (opts.tok_list && n_blocks > 2048) n_blocks = 2048;   // listed tokens: the workgroups stride over what the list needs
    const dim3 grid(static_cast<unsigned>(n_blocks));
    const long long* rg = reinterpret_cast<const long long*>(range);
    long long* kp = reinterpret_cast<long long*>(keys);
#define T5VQ_DIST_ARGS                                                                                        \
    x, w, xx, ee, rg, static_cast<int>(N), static_cast<int>(K), static_cast<int>(D), static_cast<int>(splits), \
        static_cast<int>(tiles_per_split), static_cast<long long>(code_offset)
    const bool d2 = v2 && D % (2 * C::BKC) == 0 && g_depth2;
    if (v2 && range && d2) {
        hipLaunchKernelGGL((dist_mfma_v2_kernel<C, true, true>), grid, dim3(256), C::LDS_BYTES, stream,
                           T5VQ_DIST_ARGS, atomic, kp, SelfArgs{}, opts.order);
    } else if (v2 && d2) {
        hipLaunchKernelGGL((dist_mfma_v2_kernel<C, false, true>), grid, dim3(256), C::LDS_BYTES, stream,
                           T5VQ_DIST_ARGS, atomic, kp);
    } else if (v2 && range) {
        hipLaunchKernelGGL((dist_mfma_v2_kernel<C, true, false>), grid, dim3(256), C::LDS_BYTES, stream,
                           T5VQ_DIST_ARGS, atomic, kp, SelfArgs{}, opts.order);
    } else if (v2) {
        hipLaunchKernelGGL((dist_mfma_v2_kernel<C, false, false>), grid, dim3(256), C::LDS_BYTES, stream,
                           T5VQ_DIST_ARGS, atomic, kp);
    } else if (opts.tok_list && range) {  // listed tokens only, one code tile per workgroup, merged with atomicMin
        hipLaunchKernelGGL((dist_mfma_kernel<C, true, false, false, true>), grid, dim3(256), C::LDS_BYTES, stream,
                           T5VQ_DIST_ARGS, 1, kp, dist, opts.tok_list, opts.n_listed);
    } else if (opts.tok_list) {
        hipLaunchKernelGGL((dist_mfma_kernel<C, false, false, false, true>), grid, dim3(256), C::LDS_BYTES, stream,
                           T5VQ_DIST_ARGS, 1, kp, dist, opts.tok_list, opts.n_listed);
    } else if (opts.bounded) {  // top-k pass: `range` carries one excluded key per token
        hipLaunchKernelGGL((dist_mfma_kernel<C, false, false, true>), grid, dim3(256), C::LDS_BYTES, stream,
                           T5VQ_DIST_ARGS, atomic, kp, dist);
    } else if (dist) {
        hipLaunchKernelGGL((dist_mfma_kernel<C, false, true>), grid, dim3(256), C::LDS_BYTES, stream,
                           T5VQ_DIST_ARGS, 0, kp, dist);
    } else if (range) {
        hipLaunchKernelGGL((dist_mfma_kernel<C, true, false>), grid, dim3(256), C::LDS_BYTES, stream,
                           T5VQ_DIST_ARGS, atomic, kp, dist);
    } else {
        hipLaunchKernelGGL((dist_mfma_kernel<C, false, false>), grid, dim3(256), C::LDS_BYTES, stream,
                           T5VQ_DIST_ARGS, atomic, kp, dist);
    }
#undef T5VQ_DIST_ARGS
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int launch_dist(const float* x, const float* w, const float* xx, const float* ee, const int64_t* range,
                       int64_t N, int64_t K, int64_t D, int64_t code_offset, int merge, int64_t* keys, float* dist,
                       hipStream_t stream, const DistOpts& opts = DistOpts()) {
    T5VQ_REQUIRE(x && w && xx && ee, "null input");
    T5VQ_REQUIRE(N > 0 && K > 0, "N and K must be positive");
    T5VQ_REQUIRE(D > 0 && D % 4 == 0, "D must be a positive multiple of 4");
    T5VQ_REQUIRE(aligned16(x) && aligned16(w) && aligned16(ee), "x, w, ee must be 16-byte aligned");
    T5VQ_REQUIRE(N < (1LL << 31) - TOK_TILE && K < (1LL << 31) - 256, "N or K too large");
    T5VQ_REQUIRE(code_offset >= 0 && code_offset + K < (1LL << 31), "code_offset out of range");
#define T5VQ_GO(CFG) \
    return launch_dist_cfg<CFG>(x, w, xx, ee, range, N, K, D, code_offset, merge, keys, dist, stream, opts)
#define T5VQ_GO2(CFG) \
    return launch_dist_cfg<CFG, true>(x, w, xx, ee, range, N, K, D, code_offset, merge, keys, dist, stream, opts)
    switch (g_dist_variant) {
#ifdef T5VQ_TUNING
        case 1: T5VQ_GO(Cfg1);
        case 2: T5VQ_GO(Cfg2);
        case 3: T5VQ_GO(Cfg3);
        case 4: T5VQ_GO(Cfg4);
        case 5: T5VQ_GO(Cfg5);
        case 6: T5VQ_GO(Cfg6);
        case 7: T5VQ_GO(Cfg7);
        case 8: T5VQ_GO(Cfg8);
        case 9: T5VQ_GO(Cfg9);
        case 10: T5VQ_GO(Cfg10);
        case 11: T5VQ_GO(Cfg11);
        case 12: T5VQ_GO(Cfg12);
        case 13: T5VQ_GO(Cfg13);
        case 20: T5VQ_GO2(CfgDefault);
        case 24: T5VQ_GO2(Cfg14);
        case 25: T5VQ_GO2(Cfg15);
        case 21: T5VQ_GO2(Cfg2);
        case 22: T5VQ_GO2(Cfg7);
        case 23: T5VQ_GO2(Cfg6);
#endif
        case 99: T5VQ_GO(CfgDefault);   // v1 main loop (reference point for tools/tune_dist.py)
        case 98: T5VQ_GO2(CfgDefault);  // 128-code tile regardless of the problem size
        case 97: T5VQ_GO2(CfgSmall);
        default:
            // fewer than one 128 x 128 tile per CU: halve the code tile to double the workgroups (measured:
            // C2 52 -> 33 us; at >= 256 tiles the 128-code tile wins again, e.g. C1 75 % vs 54 % of peak)
            if (!dist && ((N + TOK_TILE - 1) / TOK_TILE) * ((K + 127) / 128) < 256) T5VQ_GO2(CfgSmall);
            T5VQ_GO2(CfgDefault);
    }
#undef T5VQ_GO
#undef T5VQ_GO2
}

// ---- the self-contained form of the search for small problems (dist_mfma_v2_kernel<CfgSmall, .., SELF>) ------------
// Applies when the problem has fewer than 256 tiles of 128 x 128 (the rule that selects the 128 x 64 tile), the search
// is plain (no ranges), D % 256 == 0 and the tuning knobs are at their defaults.  With it a composite forward is
// [distance + argmin] -> [gather + statistics] with nothing in front: C2 (N = 4096, K = 512) saves the row-norm /
// preparation launch, its 4 us and the gap behind it.
constexpr int64_t BUCKET_MIN_TOKENS = 1024;     // ranged search: from this many tokens on they are visited role by role
constexpr int64_t SELF_MAX_KEY_SLOTS = 65536;   // splits x N never exceeds this for a problem the rule admits

bool self_shape(int64_t N, int64_t K) {
    return ((N + TOK_TILE - 1) / TOK_TILE) * ((K + 127) / 128) < 256;
}
bool self_applicable(int64_t N, int64_t K, int64_t D, const int64_t* range) {
    return !range && D % 256 == 0 && self_shape(N, K) && g_dist_variant == 0 && g_force_splits == 0 && g_depth2;
}

int launch_dist_self(const float* x, const float* w, const float* xx, const float* ee, int64_t N, int64_t K, int64_t D,
                     int64_t* key_slots, SelfArgs self, int* splits_out, hipStream_t stream) {
    using C = CfgSmall;
    constexpr int LDS = C::LDS_BYTES + TOK_TILE * 4;   // + the row norms of the token tile
    T5VQ_REQUIRE(x && w && ee && key_slots, "null input");
    T5VQ_REQUIRE(aligned16(x) && aligned16(w) && aligned16(ee), "x, w, ee must be 16-byte aligned");
    for (int r = 0; r < 2; ++r)
        T5VQ_REQUIRE(self.nz[r] % 4 == 0 && (self.nz[r] == 0 || (self.zero[r] && aligned16(self.zero[r]))),
                     "statistics buffers must be 16-byte aligned and a multiple of 4 floats long");
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    T5VQ_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_done[dev].load(std::memory_order_acquire)) {
        T5VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dist_mfma_v2_kernel<C, false, true, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_done[dev].store(true, std::memory_order_release);
    }
    const int64_t tok_tiles = (N + TOK_TILE - 1) / TOK_TILE;
    const int64_t code_tiles = (K + C::CODE_TILE - 1) / C::CODE_TILE;
    const int64_t splits = plan_splits(N, K, D, C::CODE_TILE, nullptr);
    T5VQ_REQUIRE(splits * N <= SELF_MAX_KEY_SLOTS, "internal: more key slots than the workspace holds");
    const int64_t tiles_per_split = (code_tiles + splits - 1) / splits;
    const dim3 grid(static_cast<unsigned>((tok_tiles + 7) / 8 * 8 * splits));
    hipLaunchKernelGGL((dist_mfma_v2_kernel<C, false, true, true>), grid, dim3(256), LDS, stream, x, w, xx, ee,
                       static_cast<const long long*>(nullptr), static_cast<int>(N), static_cast<int>(K),
                       static_cast<int>(D), static_cast<int>(splits), static_cast<int>(tiles_per_split), 0LL, 0,
                       reinterpret_cast<long long*>(key_slots), self);
    T5VQ_CHECK_HIP(hipGetLastError());
    *splits_out = static_cast<int>(splits);
    return T5VQ_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int t5vq_version(void) { return 310; }
int t5vq_set_tuning(int variant) {
    g_depth2 = variant >= 100000 ? 0 : 1;   // +100000: single staging register set
    variant %= 100000;
    g_force_splits = variant / 1000;
    g_dist_variant = variant % 1000;
    return T5VQ_OK;
}
const char* t5vq_last_error(void) { return g_err; }
int64_t t5vq_padded_codes(int64_t K) { return (K + T5VQ_CODE_PAD - 1) / T5VQ_CODE_PAD * T5VQ_CODE_PAD; }
// loss terms of the plain (unordered) statistics walk: one per token below GS_SMALL_N tokens, one per 64-token workgroup above
int64_t t5vq_gather_blocks(int64_t N) { return N < GS_SMALL_N ? N : (N + GS_TOK - 1) / GS_TOK; }
static int64_t gather_grid(int64_t N) { const int64_t t = gs_tok(N); return (N + t - 1) / t; }

static int launch_row_sqnorms(const float* x, int64_t N, float* xx, const float* w, int64_t K, float* ee, int64_t D,
                              PrepArgs prep, hipStream_t stream) {
    T5VQ_REQUIRE(D > 0 && D % 4 == 0, "D must be a positive multiple of 4");
    T5VQ_REQUIRE(N >= 0 && K >= 0, "negative row count");
    T5VQ_REQUIRE(N == 0 || (x && xx && aligned16(x)), "x/xx null or x not 16-byte aligned");
    T5VQ_REQUIRE(K == 0 || (w && ee && aligned16(w)), "w/ee null or w not 16-byte aligned");
    const int64_t kpad = t5vq_padded_codes(K);
    const int64_t bx = (N + 3) / 4, bw = (kpad + 3) / 4;
    const int64_t prep_elems = (prep.n_a + prep.n_b + prep.n_c) / 4 + prep.n_keys;
    prep.blocks = 0;
    if (prep.zero_a || prep.zero_b || prep.zero_c || prep.keys) {
        int64_t pb = (prep_elems + 1023) / 1024;  // ~4 stores per thread
        prep.blocks = static_cast<int>(pb < 1 ? 1 : (pb > 2048 ? 2048 : pb));
    }
    if (bx + bw + prep.blocks == 0) return T5VQ_OK;
    T5VQ_REQUIRE(bx + bw + prep.blocks < (1LL << 31), "too many rows");
    hipLaunchKernelGGL(row_sqnorms_kernel, dim3(static_cast<unsigned>(bx + bw + prep.blocks)), dim3(256), 0, stream, x,
                       N, xx, w, K, kpad, ee, static_cast<int>(D), bx, bx + bw, prep);
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int t5vq_row_sqnorms_f32(const float* x, int64_t N, float* xx, const float* w, int64_t K, float* ee, int64_t D,
                         t5vq_stream_t stream) {
    return launch_row_sqnorms(x, N, xx, w, K, ee, D, PrepArgs{}, static_cast<hipStream_t>(stream));
}

int t5vq_dist_argmin_f32(const float* x, const float* w, const float* xx, const float* ee, const int64_t* range,
                         const int32_t* order, int64_t N, int64_t K, int64_t D, int64_t code_offset, int merge,
                         int64_t* keys, t5vq_stream_t stream) {
    T5VQ_REQUIRE(keys, "keys is null");
    T5VQ_REQUIRE(!order || range, "order is an ordering for the ranged search: range is null");
    DistOpts opts;
    opts.order = order;
    return launch_dist(x, w, xx, ee, range, N, K, D, code_offset, merge, keys, nullptr,
                       static_cast<hipStream_t>(stream), opts);
}

int t5vq_distances_f32(const float* x, const float* w, const float* xx, const float* ee, int64_t N, int64_t K,
                       int64_t D, float* dist, t5vq_stream_t stream) {
    T5VQ_REQUIRE(dist, "dist is null");
    return launch_dist(x, w, xx, ee, nullptr, N, K, D, 0, 0, nullptr, dist, static_cast<hipStream_t>(stream));
}

int t5vq_unpack_keys(const int64_t* keys, int64_t N, int64_t* idx, float* dmin, t5vq_stream_t stream) {
    T5VQ_REQUIRE(keys && N >= 0, "keys null or N negative");
    if (N == 0) return T5VQ_OK;
    hipLaunchKernelGGL(unpack_keys_kernel, dim3(static_cast<unsigned>((N + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const long long*>(keys),
                       static_cast<long long>(N), reinterpret_cast<long long*>(idx), dmin);
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

// stats_zeroed: counts / embed_sum were zeroed by the row-norm launch's preparation tail.
// (Summing the loss partials in the last workgroup to finish -- ticket + device-scope fences -- was measured and
// rejected: the fences write back L2 and cost 5-80 us, more than the 4 us reduce launch they would replace.)
static int launch_gather(const float* x, const float* w, const int64_t* keys, int64_t N, int64_t K, int64_t D,
                         int loss_kind, float* out_ste, float* out_q, float* out_diff, int64_t* idx_out,
                         int64_t stats_begin, int64_t stats_count, float* counts, float* embed_sum,
                         float* loss_partial, bool stats_zeroed, const int32_t* perm, hipStream_t stream,
                         int key_sets = 1) {
    T5VQ_REQUIRE(x && w && keys, "null input");
    T5VQ_REQUIRE(N > 0 && K > 0 && N < (1LL << 31) && K < (1LL << 31), "N, K out of range");
    T5VQ_REQUIRE(D > 0 && D % 4 == 0, "D must be a positive multiple of 4");
    T5VQ_REQUIRE(aligned16(x) && aligned16(w), "x, w must be 16-byte aligned");
    T5VQ_REQUIRE((!out_ste || aligned16(out_ste)) && (!out_q || aligned16(out_q)) && (!out_diff || aligned16(out_diff)),
                 "outputs must be 16-byte aligned");
    T5VQ_REQUIRE(loss_kind == T5VQ_LOSS_MSE || loss_kind == T5VQ_LOSS_COSINE, "unknown loss_kind");
    const int nch = pick_nch(D);
    T5VQ_REQUIRE(nch > 0, "D > 2048 is not supported");
    T5VQ_REQUIRE(stats_begin >= 0 && stats_count >= 0 && stats_begin + stats_count <= K, "stats slice outside [0,K)");
    T5VQ_REQUIRE(!embed_sum || aligned16(embed_sum), "embed_sum must be 16-byte aligned");
    if (!stats_zeroed) {
        if (counts && stats_count)
            T5VQ_CHECK_HIP(hipMemsetAsync(counts, 0, static_cast<size_t>(stats_count) * 4, stream));
        if (embed_sum && stats_count)
            T5VQ_CHECK_HIP(hipMemsetAsync(embed_sum, 0, static_cast<size_t>(stats_count) * D * 4, stream));
    }
    const dim3 grid(static_cast<unsigned>(gather_grid(N)));
    const size_t lds = static_cast<size_t>(4) * nch * 256 * 4;
    const long long* kp = reinterpret_cast<const long long*>(keys);
    long long* ip = reinterpret_cast<long long*>(idx_out);
#define T5VQ_GS_T(NCH_, COS_, TOK_)                                                                                 \
    hipLaunchKernelGGL((gather_stats_kernel<NCH_, COS_, TOK_>), grid, dim3(256), lds, stream, x, w, kp,             \
                       static_cast<int>(N), static_cast<int>(K), static_cast<int>(D), out_ste, out_q, out_diff, ip, \
                       static_cast<int>(stats_begin), static_cast<int>(stats_count), counts, embed_sum, loss_partial, \
                       perm, key_sets)
#define T5VQ_GS(NCH_, COS_)                                   \
    if (gs_tok(N) == GS_TOK_TINY) {                           \
        T5VQ_GS_T(NCH_, COS_, GS_TOK_TINY);                   \
    } else if (gs_tok(N) == GS_TOK_SMALL) {                   \
        T5VQ_GS_T(NCH_, COS_, GS_TOK_SMALL);                  \
    } else {                                                  \
        T5VQ_GS_T(NCH_, COS_, GS_TOK);                        \
    }
#define T5VQ_GS_D(COS_)                \
    switch (nch) {                     \
        case 1: T5VQ_GS(1, COS_); break; \
        case 2: T5VQ_GS(2, COS_); break; \
        case 3: T5VQ_GS(3, COS_); break; \
        case 4: T5VQ_GS(4, COS_); break; \
        default: T5VQ_GS(8, COS_); break; \
    }
    if (loss_kind == T5VQ_LOSS_COSINE) {
        T5VQ_GS_D(true)
    } else {
        T5VQ_GS_D(false)
    }
#undef T5VQ_GS_D
#undef T5VQ_GS
#undef T5VQ_GS_T
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int t5vq_gather_stats_f32(const float* x, const float* w, const int64_t* keys, int64_t N, int64_t K, int64_t D,
                          int loss_kind, float* out_ste, float* out_q, float* out_diff, int64_t* idx_out,
                          int64_t stats_begin, int64_t stats_count, float* counts, float* embed_sum,
                          float* loss_partial, const int32_t* perm, t5vq_stream_t stream) {
    return launch_gather(x, w, keys, N, K, D, loss_kind, out_ste, out_q, out_diff, idx_out, stats_begin, stats_count,
                         counts, embed_sum, loss_partial, false, perm, static_cast<hipStream_t>(stream));
}

static int order_by_code(const int64_t* keys, int64_t N, int64_t K, int32_t* perm, int32_t* work, bool work_zeroed,
                         hipStream_t stream);
// exclusive scan of work[0, K) by one workgroup; the counters-per-thread form that covers K
static void launch_code_scan(int32_t* work, int64_t N, int64_t K, hipStream_t stream) {
    const int64_t per = (K + 1023) / 1024;
#define T5VQ_SCAN(PER_) \
    hipLaunchKernelGGL(code_scan_kernel<PER_>, dim3(1), dim3(1024), 0, stream, work, static_cast<int>(N), static_cast<int>(K))
    if (per <= 1) { T5VQ_SCAN(1); }
    else if (per <= 2) { T5VQ_SCAN(2); }
    else if (per <= 4) { T5VQ_SCAN(4); }
    else if (per <= 8) { T5VQ_SCAN(8); }
    else if (per <= 16) { T5VQ_SCAN(16); }
    else if (per <= 32) { T5VQ_SCAN(32); }
    else if (per <= 64) { T5VQ_SCAN(64); }
    else { T5VQ_SCAN(0); }
#undef T5VQ_SCAN
}
int t5vq_order_by_code(const int64_t* keys, int64_t N, int64_t K, int32_t* perm, int32_t* work,
                       t5vq_stream_t stream_) {
    return order_by_code(keys, N, K, perm, work, false, static_cast<hipStream_t>(stream_));
}

static int order_by_code(const int64_t* keys, int64_t N, int64_t K, int32_t* perm, int32_t* work, bool work_zeroed,
                         hipStream_t stream) {
    T5VQ_REQUIRE(keys && perm && work, "null pointer");
    T5VQ_REQUIRE(N > 0 && K > 0 && N < (1LL << 31) && K < (1LL << 31), "N, K out of range");
    const long long* kp = reinterpret_cast<const long long*>(keys);
    if (!work_zeroed) T5VQ_CHECK_HIP(hipMemsetAsync(work, 0, static_cast<size_t>(K + 2) * 4, stream));
    const dim3 grid(static_cast<unsigned>((N + 255) / 256));
    hipLaunchKernelGGL(code_hist_kernel, grid, dim3(256), 0, stream, kp, static_cast<int>(N), static_cast<int>(K), work);
    launch_code_scan(work, N, K, stream);
    hipLaunchKernelGGL(code_place_kernel, grid, dim3(256), 0, stream, kp, static_cast<int>(N), static_cast<int>(K), work,
                       perm);
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

// Bucket granularity: 128 codes (the scan's code tile), doubled until the T (T + 1) + 1 buckets and the scan kernel's two
// trailing words fit the K + 8 words every caller of the counting sort holds.
static RangeBins plan_range_bins(int64_t K, int64_t code_offset) {
    RangeBins rb;
    rb.code_offset = code_offset;
    rb.K = static_cast<int>(K);
    rb.shift = 7;
    for (;;) {
        rb.T = static_cast<int>((K + (1LL << rb.shift) - 1) >> rb.shift);
        if (static_cast<int64_t>(range_bin_count(rb.T)) + 2 <= K + 8) break;
        ++rb.shift;
    }
    return rb;
}

static int order_by_range(const int64_t* range, int64_t N, int64_t K, int64_t code_offset, int32_t* order,
                          int32_t* work, hipStream_t stream) {
    T5VQ_REQUIRE(range && order && work, "null pointer");
    T5VQ_REQUIRE(N > 0 && K > 0 && N < (1LL << 31) && K < (1LL << 31) - 512, "N, K out of range");
    const RangeBins rb = plan_range_bins(K, code_offset);
    const int bins = range_bin_count(rb.T);
    T5VQ_CHECK_HIP(hipMemsetAsync(work, 0, static_cast<size_t>(bins + 2) * 4, stream));
    const dim3 grid(static_cast<unsigned>((N + RANGE_WG - 1) / RANGE_WG));
    const long long* rg = reinterpret_cast<const long long*>(range);
    hipLaunchKernelGGL(range_hist_kernel, grid, dim3(RANGE_WG), 0, stream, rg, static_cast<int>(N), rb, work);
    launch_code_scan(work, N, bins, stream);
    hipLaunchKernelGGL(range_place_kernel, grid, dim3(RANGE_WG), 0, stream, rg, static_cast<int>(N), rb, work, order);
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int t5vq_order_by_range(const int64_t* range, int64_t N, int64_t K, int64_t code_offset, int32_t* order,
                        int32_t* work, t5vq_stream_t stream) {
    return order_by_range(range, N, K, code_offset, order, work, static_cast<hipStream_t>(stream));
}

int t5vq_reduce_partials_f32(const float* partials, int64_t n, float* loss_sum, t5vq_stream_t stream) {
    T5VQ_REQUIRE(partials && loss_sum && n >= 0, "null pointer or negative n");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), partials,
                       static_cast<long long>(n), loss_sum);
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int t5vq_ema_cluster_f32(float* cluster_size, const float* counts, const float* loss_sum, int64_t n_loss_terms,
                         int64_t K, int64_t n_tokens, float loss_denom, float decay, float one_minus_decay,
                         float epsilon, float k_epsilon, float commitment_cost, int plain_variant, int mode,
                         float* scalars, t5vq_stream_t stream) {
    T5VQ_REQUIRE(scalars && (counts || mode == 3), "counts/scalars null");
    T5VQ_REQUIRE(mode >= 0 && mode <= 3, "mode must be 0..3");
    T5VQ_REQUIRE(mode != 1 || cluster_size, "cluster_size null with mode 1");
    T5VQ_REQUIRE(K > 0 && K < (1LL << 31) && n_tokens > 0, "K, n_tokens out of range");
    T5VQ_REQUIRE(!loss_sum || n_loss_terms >= 1, "n_loss_terms must be at least 1 when loss_sum is given");
    const bool vec = K % 4 == 0 && (!counts || aligned16(counts)) && (!cluster_size || aligned16(cluster_size));
#define T5VQ_CL(VEC_)                                                                                              \
    hipLaunchKernelGGL((ema_cluster_kernel<VEC_>), dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream),       \
                       cluster_size, counts, loss_sum, static_cast<long long>(n_loss_terms), static_cast<int>(K),  \
                       static_cast<float>(n_tokens), loss_denom, decay, one_minus_decay, epsilon, k_epsilon,      \
                       commitment_cost, plain_variant, mode, scalars)
    if (vec) {
        T5VQ_CL(true);
    } else {
        T5VQ_CL(false);
    }
#undef T5VQ_CL
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int t5vq_ema_embed_img_f32(float* ema_w, float* w, const float* embed_sum, const float* cluster_size, int64_t K,
                           int64_t D, float decay, float one_minus_decay, float* ee_out, void* code_image,
                           t5vq_stream_t stream) {
    T5VQ_REQUIRE(ema_w && w && embed_sum && cluster_size, "null pointer");
    T5VQ_REQUIRE(K > 0 && K < (1LL << 31) && D > 0 && D % 4 == 0, "K, D out of range");
    T5VQ_REQUIRE(aligned16(ema_w) && aligned16(w) && aligned16(embed_sum), "buffers must be 16-byte aligned");
    T5VQ_REQUIRE((reinterpret_cast<uintptr_t>(code_image) & 255u) == 0, "code_image must be 256-byte aligned");
    T5VQ_REQUIRE(!code_image || D <= 2048, "the update writes a code image for D <= 2048 only");
#define T5VQ_EE(IMG_)                                                                                              \
    hipLaunchKernelGGL((ema_embed_kernel<IMG_>), dim3(static_cast<unsigned>((K + 3) / 4)), dim3(256), 0,              \
                       static_cast<hipStream_t>(stream), ema_w, w, embed_sum, cluster_size, static_cast<int>(K),     \
                       static_cast<int>(D), decay, one_minus_decay, ee_out, t5vq_code_image_parts(code_image, K, D), \
                       static_cast<int>(t5vq_padded_codes(K)))
    switch (code_image ? pick_nch(D) : 0) {   // D <= 2048 with an image (checked above)
        case 0: T5VQ_EE(0); break;
        case 1: T5VQ_EE(1); break;
        case 2: T5VQ_EE(2); break;
        case 3: T5VQ_EE(3); break;
        case 4: T5VQ_EE(4); break;
        default: T5VQ_EE(8); break;
    }
#undef T5VQ_EE
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}
int t5vq_ema_embed_f32(float* ema_w, float* w, const float* embed_sum, const float* cluster_size, int64_t K,
                       int64_t D, float decay, float one_minus_decay, float* ee_out, t5vq_stream_t stream) {
    return t5vq_ema_embed_img_f32(ema_w, w, embed_sum, cluster_size, K, D, decay, one_minus_decay, ee_out, nullptr, stream);
}

int t5vq_max_norm_f32(const float* xx, int64_t N, int32_t* max_xx_bits, t5vq_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    T5VQ_REQUIRE(xx && max_xx_bits && N > 0, "null pointer or empty input");
    T5VQ_CHECK_HIP(hipMemsetAsync(max_xx_bits, 0, 4, stream));
    int64_t blocks = (N + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(max_norm_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, xx,
                       static_cast<long long>(N), max_xx_bits);
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int t5vq_embed_sum_fixed_f32(const float* x, const int64_t* keys, const int32_t* max_xx_bits, int64_t N, int64_t K,
                             int64_t D, int64_t stats_begin, int64_t stats_count, int64_t* embed_sum,
                             t5vq_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    T5VQ_REQUIRE(x && keys && max_xx_bits && embed_sum, "null pointer");
    T5VQ_REQUIRE(N > 0 && K > 0 && N < (1LL << 24) && K < (1LL << 31), "N must be below 2^24, K below 2^31");
    T5VQ_REQUIRE(D > 0 && D % 4 == 0 && aligned16(x), "D % 4 and alignment");
    T5VQ_REQUIRE(stats_begin >= 0 && stats_count >= 0 && stats_begin + stats_count <= K, "stats slice outside [0,K)");
    const int nch = pick_nch(D);
    T5VQ_REQUIRE(nch > 0 && nch <= 4, "D > 1024 is not supported by the fixed-point statistics");
    if (stats_count == 0) return T5VQ_OK;
    T5VQ_CHECK_HIP(hipMemsetAsync(embed_sum, 0, static_cast<size_t>(stats_count) * D * 8, stream));
    const int tok = gs_tok(N);
    const dim3 grid(static_cast<unsigned>((N + tok - 1) / tok));
    const size_t lds = static_cast<size_t>(4) * nch * 256 * 8;
    const long long* kp = reinterpret_cast<const long long*>(keys);
    long long* ep = reinterpret_cast<long long*>(embed_sum);
#define T5VQ_FX(NCH_, TOK_)                                                                                      \
    hipLaunchKernelGGL((embed_sum_fixed_kernel<NCH_, TOK_>), grid, dim3(256), lds, stream, x, kp, static_cast<int>(N), \
                       static_cast<int>(K), static_cast<int>(D), max_xx_bits, static_cast<int>(stats_begin),        \
                       static_cast<int>(stats_count), ep)
#define T5VQ_FX_T(NCH_)                      \
    if (tok == GS_TOK_TINY) {                \
        T5VQ_FX(NCH_, GS_TOK_TINY);          \
    } else if (tok == GS_TOK_SMALL) {        \
        T5VQ_FX(NCH_, GS_TOK_SMALL);         \
    } else {                                 \
        T5VQ_FX(NCH_, GS_TOK);               \
    }
    switch (nch) {
        case 1: T5VQ_FX_T(1); break;
        case 2: T5VQ_FX_T(2); break;
        case 3: T5VQ_FX_T(3); break;
        default: T5VQ_FX_T(4); break;
    }
#undef T5VQ_FX_T
#undef T5VQ_FX
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int t5vq_ema_embed_fixed_img_f32(float* ema_w, float* w, const int64_t* embed_sum, const int32_t* max_xx_bits,
                                 const float* cluster_size, int64_t K, int64_t D, float decay, float one_minus_decay,
                                 float* ee_out, void* code_image, t5vq_stream_t stream) {
    T5VQ_REQUIRE(ema_w && w && embed_sum && max_xx_bits && cluster_size, "null pointer");
    T5VQ_REQUIRE(K > 0 && K < (1LL << 31) && D > 0 && D % 4 == 0, "K, D out of range");
    T5VQ_REQUIRE(!code_image || (aligned16(w) && (reinterpret_cast<uintptr_t>(code_image) & 255u) == 0),
                 "code_image must be 256-byte aligned (and w 16-byte aligned)");
    T5VQ_REQUIRE(!code_image || D <= 2048, "the update writes a code image for D <= 2048 only");
#define T5VQ_EF(IMG_)                                                                                               \
    hipLaunchKernelGGL((ema_embed_fixed_kernel<IMG_>), dim3(static_cast<unsigned>((K + 3) / 4)), dim3(256), 0,         \
                       static_cast<hipStream_t>(stream), ema_w, w, reinterpret_cast<const long long*>(embed_sum),     \
                       max_xx_bits, cluster_size, static_cast<int>(K), static_cast<int>(D), decay, one_minus_decay,   \
                       ee_out, t5vq_code_image_parts(code_image, K, D), static_cast<int>(t5vq_padded_codes(K)))
    switch (code_image ? pick_nch(D) : 0) {
        case 0: T5VQ_EF(0); break;
        case 1: T5VQ_EF(1); break;
        case 2: T5VQ_EF(2); break;
        case 3: T5VQ_EF(3); break;
        case 4: T5VQ_EF(4); break;
        default: T5VQ_EF(8); break;
    }
#undef T5VQ_EF
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}
int t5vq_ema_embed_fixed_f32(float* ema_w, float* w, const int64_t* embed_sum, const int32_t* max_xx_bits,
                             const float* cluster_size, int64_t K, int64_t D, float decay, float one_minus_decay,
                             float* ee_out, t5vq_stream_t stream) {
    return t5vq_ema_embed_fixed_img_f32(ema_w, w, embed_sum, max_xx_bits, cluster_size, K, D, decay, one_minus_decay,
                                        ee_out, nullptr, stream);
}

int t5vq_ste_backward_f32(const float* g_out, const float* diff, const float* g_loss, float coef, int64_t n_elems,
                          float* grad_x, t5vq_stream_t stream) {
    T5VQ_REQUIRE(diff && grad_x, "null pointer");
    T5VQ_REQUIRE(n_elems > 0 && n_elems % 4 == 0, "n_elems must be a positive multiple of 4");
    T5VQ_REQUIRE(aligned16(diff) && aligned16(grad_x) && (!g_out || aligned16(g_out)), "16-byte alignment");
    const long long n4 = n_elems / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ste_backward_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), g_out, diff, g_loss, coef, n4, grad_x);
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int t5vq_scatter_rows_f32(const float* rows, const int64_t* idx, const float* scale_dev, float scale_host, int64_t N,
                          int64_t K, int64_t D, float* out, t5vq_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    T5VQ_REQUIRE(rows && idx && out, "null pointer");
    T5VQ_REQUIRE(N > 0 && K > 0 && N < (1LL << 31) && K < (1LL << 31), "N, K out of range");
    T5VQ_REQUIRE(D > 0 && D % 4 == 0 && aligned16(rows), "D % 4 and alignment");
    const int nch = pick_nch(D);
    T5VQ_REQUIRE(nch > 0, "D > 2048 is not supported");
    T5VQ_CHECK_HIP(hipMemsetAsync(out, 0, static_cast<size_t>(K) * D * 4, stream));
    const dim3 grid(static_cast<unsigned>((N + GS_TOK - 1) / GS_TOK));
    const size_t lds = static_cast<size_t>(4) * nch * 256 * 4;
    const long long* ip = reinterpret_cast<const long long*>(idx);
#define T5VQ_SC(NCH_)                                                                                            \
    hipLaunchKernelGGL((scatter_rows_kernel<NCH_>), grid, dim3(256), lds, stream, rows, ip, scale_dev, scale_host, \
                       static_cast<int>(N), static_cast<int>(K), static_cast<int>(D), out)
    switch (nch) {
        case 1: T5VQ_SC(1); break;
        case 2: T5VQ_SC(2); break;
        case 3: T5VQ_SC(3); break;
        case 4: T5VQ_SC(4); break;
        default: T5VQ_SC(8); break;
    }
#undef T5VQ_SC
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int t5vq_cosine_backward_f32(const float* x, const float* q, const float* g_out, const float* g_loss, float coef_x,
                             float coef_q, int64_t N, int64_t D, float* grad_x, float* grad_q,
                             t5vq_stream_t stream) {
    T5VQ_REQUIRE(x && q && grad_x, "null pointer");
    T5VQ_REQUIRE(N > 0 && N < (1LL << 31) && D > 0 && D % 4 == 0, "N, D out of range");
    T5VQ_REQUIRE(aligned16(x) && aligned16(q) && aligned16(grad_x) && (!grad_q || aligned16(grad_q)) &&
                     (!g_out || aligned16(g_out)), "16-byte alignment");
    hipLaunchKernelGGL(cosine_backward_kernel, dim3(static_cast<unsigned>((N + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, q, g_out, g_loss, coef_x, coef_q, static_cast<int>(N),
                       static_cast<int>(D), grad_x, grad_q);
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int t5vq_one_hot_f32(const int64_t* idx, int64_t N, int64_t K, float* enc, t5vq_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    T5VQ_REQUIRE(idx && enc && N > 0 && K > 0, "null pointer or empty shape");
    T5VQ_CHECK_HIP(hipMemsetAsync(enc, 0, static_cast<size_t>(N) * K * 4, stream));
    hipLaunchKernelGGL(one_hot_kernel, dim3(static_cast<unsigned>((N + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const long long*>(idx), static_cast<long long>(N), static_cast<long long>(K),
                       enc);
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int64_t t5vq_rmsnorm_backward_blocks(int64_t N) {
    const int64_t b = (N + 3) / 4;
    return b < 1 ? 1 : (b > 1024 ? 1024 : b);
}

int t5vq_rmsnorm_sqnorms_f32(const float* hidden, const float* gamma, const uint8_t* keep, float scale, float eps,
                             int64_t N, int64_t D, float* x_out, float* xx, float* rms, t5vq_stream_t stream) {
    return t5vq_rmsnorm_sqnorms_img_f32(hidden, gamma, keep, scale, eps, N, D, x_out, xx, rms, nullptr, stream);
}
int t5vq_rmsnorm_sqnorms_img_f32(const float* hidden, const float* gamma, const uint8_t* keep, float scale, float eps,
                                 int64_t N, int64_t D, float* x_out, float* xx, float* rms, void* token_image,
                                 t5vq_stream_t stream) {
    T5VQ_REQUIRE(hidden && gamma && x_out, "null pointer");
    T5VQ_REQUIRE((reinterpret_cast<uintptr_t>(token_image) & 255u) == 0, "token_image must be 256-byte aligned");
    const T5vqTokenImage timg = t5vq_token_image_parts(token_image, N, D);
    T5VQ_REQUIRE(N > 0 && N < (1LL << 31) && D > 0 && D % 4 == 0, "N, D out of range");
    T5VQ_REQUIRE(aligned16(hidden) && aligned16(gamma) && aligned16(x_out), "16-byte alignment");
    T5VQ_REQUIRE(!keep || (reinterpret_cast<uintptr_t>(keep) & 3u) == 0, "keep mask must be 4-byte aligned");
    const int nch = pick_nch(D);
    T5VQ_REQUIRE(nch > 0, "D > 2048 is not supported");
    const dim3 grid(static_cast<unsigned>((N + 3) / 4));
#define T5VQ_RN(NCH_)                                                                                              \
    hipLaunchKernelGGL((rmsnorm_sqnorms_kernel<NCH_>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), hidden, \
                       gamma, keep, scale, eps, static_cast<int>(N), static_cast<int>(D), x_out, xx, rms, timg)
    switch (nch) {
        case 1: T5VQ_RN(1); break;
        case 2: T5VQ_RN(2); break;
        case 3: T5VQ_RN(3); break;
        case 4: T5VQ_RN(4); break;
        default: T5VQ_RN(8); break;
    }
#undef T5VQ_RN
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

int t5vq_rmsnorm_backward_f32(const float* grad_x, const float* hidden, const float* gamma, const uint8_t* keep,
                              float scale, const float* rms, int64_t N, int64_t D, float* grad_hidden,
                              float* grad_gamma, float* workspace, t5vq_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    T5VQ_REQUIRE(grad_x && hidden && gamma && rms && grad_hidden, "null pointer");
    T5VQ_REQUIRE(!grad_gamma || workspace, "grad_gamma needs the [t5vq_rmsnorm_backward_blocks(N), D] workspace");
    T5VQ_REQUIRE(N > 0 && N < (1LL << 31) && D > 0 && D % 4 == 0, "N, D out of range");
    T5VQ_REQUIRE(aligned16(grad_x) && aligned16(hidden) && aligned16(gamma) && aligned16(grad_hidden) &&
                     (!workspace || aligned16(workspace)), "16-byte alignment");
    T5VQ_REQUIRE(!keep || (reinterpret_cast<uintptr_t>(keep) & 3u) == 0, "keep mask must be 4-byte aligned");
    const int nch = pick_nch(D);
    T5VQ_REQUIRE(nch > 0, "D > 2048 is not supported");
    const int64_t blocks = t5vq_rmsnorm_backward_blocks(N);
    const dim3 grid(static_cast<unsigned>(blocks));
    const size_t lds = static_cast<size_t>(4) * nch * 256 * 4;
    float* part = grad_gamma ? workspace : nullptr;
#define T5VQ_RB(NCH_)                                                                                           \
    hipLaunchKernelGGL((rmsnorm_backward_kernel<NCH_>), grid, dim3(256), lds, stream, grad_x, hidden, gamma, keep, \
                       scale, rms, static_cast<int>(N), static_cast<int>(D), grad_hidden, part)
    switch (nch) {
        case 1: T5VQ_RB(1); break;
        case 2: T5VQ_RB(2); break;
        case 3: T5VQ_RB(3); break;
        case 4: T5VQ_RB(4); break;
        default: T5VQ_RB(8); break;
    }
#undef T5VQ_RB
    T5VQ_CHECK_HIP(hipGetLastError());
    if (grad_gamma) {
        hipLaunchKernelGGL(rmsnorm_gamma_reduce_kernel, dim3(static_cast<unsigned>((D + 255) / 256)), dim3(256), 0,
                           stream, part, static_cast<int>(blocks), static_cast<int>(D), grad_gamma);
        T5VQ_CHECK_HIP(hipGetLastError());
    }
    return T5VQ_OK;
}

// Internal (t5vq_internal.h): fp32 scan of the tokens in tok_list[0 .. *n_listed) only; their keys must hold KEY_NONE.
int t5vq_internal_scan_listed(const float* x, const float* w, const float* xx, const float* ee, const int64_t* range,
                              int64_t N, int64_t K, int64_t D, int64_t code_offset, const int* tok_list,
                              const int* n_listed, int64_t* keys, hipStream_t stream) {
    DistOpts opts;
    opts.tok_list = tok_list;
    opts.n_listed = n_listed;
    return launch_dist_cfg<CfgDefault, false>(x, w, xx, ee, range, N, K, D, code_offset, 1, keys, nullptr, stream, opts);
}

int t5vq_dist_topk_f32(const float* x, const float* w, const float* xx, const float* ee, int64_t N, int64_t K,
                       int64_t D, int64_t topk, int64_t* keys, t5vq_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    T5VQ_REQUIRE(keys, "keys is null");
    T5VQ_REQUIRE(topk >= 1 && topk <= K, "topk must be in [1, K]");
    // pass 0: the plain fused argmin; pass j > 0: the same search restricted to pairs after the pass j-1 winner
    int rc = launch_dist(x, w, xx, ee, nullptr, N, K, D, 0, 0, keys, nullptr, stream);
    for (int64_t j = 1; j < topk && rc == T5VQ_OK; ++j) {
        DistOpts opts;
        opts.bounded = true;
        rc = launch_dist(x, w, xx, ee, keys + (j - 1) * N, N, K, D, 0, 0, keys + j * N, nullptr, stream, opts);
    }
    return rc;
}

int t5vq_blend_argmin_f32(const float* dist_left, const float* dist_right, float weight_left, float weight_right,
                          int64_t N, int64_t K, int64_t* keys, t5vq_stream_t stream) {
    T5VQ_REQUIRE(dist_left && dist_right && keys, "null pointer");
    T5VQ_REQUIRE(N > 0 && K > 0 && N < (1LL << 31) && K < (1LL << 31), "N, K out of range");
    hipLaunchKernelGGL(blend_argmin_kernel, dim3(static_cast<unsigned>((N + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dist_left, dist_right, weight_left, weight_right,
                       static_cast<int>(N), static_cast<int>(K), reinterpret_cast<long long*>(keys));
    T5VQ_CHECK_HIP(hipGetLastError());
    return T5VQ_OK;
}

// ---- composite entry points: the whole forward / the whole finalize behind ONE host call each -----------------
namespace {
inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
struct FwdWorkspace {
    int64_t keys, xx, ee, partial, perm, work, total;
    FwdWorkspace(int64_t N, int64_t K) {
        keys = 0;                                                   // int64 [N]; small problems: [splits][N] key slots
        xx = align_up(8 * (self_shape(N, K) ? SELF_MAX_KEY_SLOTS : N), 16);   // fp32 [N]
        ee = align_up(xx + 4 * N, 16);                              // fp32 [padded(K)], 16-byte aligned
        partial = align_up(ee + 4 * t5vq_padded_codes(K), 16);      // fp32 [gather_blocks(N)], [N] with ordered tokens
        perm = align_up(partial + 4 * N, 16);                       // int32 [N]: tokens in role order for the ranged
                                                                    // search, then (rewritten) ordered by code
        work = align_up(perm + 4 * N, 16);                          // int32 [K + 8]: counting-sort cursors
        total = align_up(work + 4 * (K + 8), 256);
    }
};
}  // namespace

int64_t t5vq_forward_workspace_bytes(int64_t N, int64_t K) {
    if (N < 0 || K < 0) return 0;
    return FwdWorkspace(N, K).total;
}

// Launches: [row norms + buffer preparation] -> [distance + argmin] -> ([order tokens by code: 3 small launches],
// training with N >= 32768) -> [gather + statistics] -> [loss sum].
int64_t t5vq_forward_loss_terms(int64_t N, int with_embed_sum) {
    if (N <= 0) return 0;
    return with_embed_sum ? N : t5vq_gather_blocks(N);   // ordered walk (N >= GS_SMALL_N) or small batch: one per token
}

static int vq_forward_impl(const float* x, const float* w, const int64_t* range, const float* xx_in,
                           const float* ee_in, int64_t N, int64_t K, int64_t D, int loss_kind, void* workspace,
                           float* out_ste, float* out_q, float* out_diff, int64_t* idx_out, float* counts,
                           float* embed_sum, float* loss_terms, float* loss_sum, hipStream_t stream) {
    const FwdWorkspace ws(N, K);
    T5VQ_REQUIRE(x && w, "null input");
    T5VQ_REQUIRE(!ee_in || aligned16(ee_in), "ee_in must be 16-byte aligned");
    char* base = static_cast<char*>(workspace);
    int64_t* keys = reinterpret_cast<int64_t*>(base + ws.keys);
    const float* xx = xx_in ? xx_in : reinterpret_cast<float*>(base + ws.xx);
    const float* ee = ee_in ? ee_in : reinterpret_cast<float*>(base + ws.ee);
    float* partial = loss_terms ? loss_terms : (loss_sum ? reinterpret_cast<float*>(base + ws.partial) : nullptr);
    const bool order = embed_sum && N >= GS_SMALL_N;   // statistics wanted, enough tokens for the ordering to pay
    if (self_applicable(N, K, D, range) && !order && K % 4 == 0 && (K * D) % 4 == 0 &&
        (!counts || aligned16(counts)) && (!embed_sum || aligned16(embed_sum))) {
        // small problem: [|e|^2 only if it was not given] -> [self-contained distance + argmin] -> [gather + statistics]
        int rc = T5VQ_OK;
        if (!ee_in) {
            rc = launch_row_sqnorms(nullptr, 0, nullptr, w, K, reinterpret_cast<float*>(base + ws.ee), D, PrepArgs{}, stream);
            if (rc != T5VQ_OK) return rc;
        }
        SelfArgs sa{};
        sa.zero[0] = counts;
        sa.nz[0] = counts ? K : 0;
        sa.zero[1] = embed_sum;
        sa.nz[1] = embed_sum ? K * D : 0;
        int splits = 1;
        rc = launch_dist_self(x, w, xx_in, ee, N, K, D, keys, sa, &splits, stream);
        if (rc != T5VQ_OK) return rc;
        rc = launch_gather(x, w, keys, N, K, D, loss_kind, out_ste, out_q, out_diff, idx_out, 0, K, counts, embed_sum,
                           partial, true, nullptr, stream, splits);
        if (rc != T5VQ_OK || !loss_sum) return rc;
        return t5vq_reduce_partials_f32(partial, t5vq_gather_blocks(N), loss_sum, stream);
    }
    // Disentangled codebook, more than a few token tiles: visit the tokens role by role (a counting sort of the
    // ranges; the order lives in the region the by-code ordering rewrites after the search).
    const int32_t* role_order = nullptr;
    if (range && N >= BUCKET_MIN_TOKENS) {
        int32_t* p = reinterpret_cast<int32_t*>(base + ws.perm);
        const int rc0 = order_by_range(range, N, K, 0, p, reinterpret_cast<int32_t*>(base + ws.work), stream);
        if (rc0 != T5VQ_OK) return rc0;
        role_order = p;
    }
    PrepArgs prep{};
    prep.zero_a = counts;
    prep.n_a = counts ? K : 0;
    prep.zero_b = embed_sum;
    prep.n_b = embed_sum ? K * D : 0;
    prep.keys = reinterpret_cast<long long*>(keys);
    prep.n_keys = N;
    if (order) {   // the counting sort's cursors are zeroed by the same launch (a memset launch less)
        prep.zero_c = reinterpret_cast<float*>(base + ws.work);
        prep.n_c = K + 2;
    }
    // |x|^2 may come from the producer of x (t5vq_rmsnorm_sqnorms_f32) and |e|^2 from the finalize call that wrote
    // the codebook (t5vq_vq_ema_finalize_f32's ee_out): whatever is given is not recomputed here
    int rc = launch_row_sqnorms(xx_in ? nullptr : x, xx_in ? 0 : N, xx_in ? nullptr : reinterpret_cast<float*>(base + ws.xx),
                                ee_in ? nullptr : w, ee_in ? 0 : K, ee_in ? nullptr : reinterpret_cast<float*>(base + ws.ee),
                                D, prep, stream);
    if (rc != T5VQ_OK) return rc;
    DistOpts opts;
    opts.keys_prefilled = true;
    opts.order = role_order;
    rc = launch_dist(x, w, xx, ee, range, N, K, D, 0, 0, keys, nullptr, stream, opts);
    if (rc != T5VQ_OK) return rc;
    const int32_t* perm = nullptr;
    if (order) {
        int32_t* p = reinterpret_cast<int32_t*>(base + ws.perm);
        rc = order_by_code(keys, N, K, p, reinterpret_cast<int32_t*>(base + ws.work), true, stream);
        if (rc != T5VQ_OK) return rc;
        perm = p;
    }
    rc = launch_gather(x, w, keys, N, K, D, loss_kind, out_ste, out_q, out_diff, idx_out, 0, K, counts, embed_sum,
                       partial, true, perm, stream);
    if (rc != T5VQ_OK || !loss_sum) return rc;
    return t5vq_reduce_partials_f32(partial, perm ? N : t5vq_gather_blocks(N), loss_sum, stream);
}

int t5vq_vq_forward_f32(const float* x, const float* w, const int64_t* range, const float* xx_in, const float* ee_in,
                        int64_t N, int64_t K, int64_t D, int loss_kind, void* workspace, int64_t workspace_bytes,
                        float* out_ste, float* out_q, float* out_diff, int64_t* idx_out, float* counts,
                        float* embed_sum, float* loss_terms, float* loss_sum, t5vq_stream_t stream_) {
    T5VQ_REQUIRE(N > 0 && K > 0, "N and K must be positive");
    T5VQ_REQUIRE(workspace && aligned16(workspace), "workspace null or not 16-byte aligned");
    T5VQ_REQUIRE(workspace_bytes >= FwdWorkspace(N, K).total, "workspace smaller than t5vq_forward_workspace_bytes(N, K)");
    return vq_forward_impl(x, w, range, xx_in, ee_in, N, K, D, loss_kind, workspace, out_ste, out_q, out_diff, idx_out,
                           counts, embed_sum, loss_terms, loss_sum, static_cast<hipStream_t>(stream_));
}

// ---- The forward for callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc): the per-call temporaries of t5vq_vq_forward_f32 (keys, |x|^2, |e|^2, loss terms, token order, sort
// cursors) live in ONE file-scope device array instead of a caller-brought workspace.  Every GPU the library runs on
// holds its own copy of the array, at its own address, until the process ends.
static __device__ __align__(256) unsigned char scratch[T5VQ_NOALLOC_SCRATCH_BYTES];

namespace {
// In host code `scratch` is only a stand-in for the symbol: the device address of each GPU's copy is looked up once
// per device (with that device current) and kept here; only these pointers ever reach a kernel, memset or copy.
std::mutex g_scratch_mutex;
void* g_scratch_ptr[64] = {nullptr};

int scratch_for_current_device(void** out) {
    int dev = 0;
    T5VQ_CHECK_HIP(hipGetDevice(&dev));
    T5VQ_REQUIRE(dev >= 0 && dev < 64, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        T5VQ_CHECK_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
        T5VQ_REQUIRE(ptr && (reinterpret_cast<uintptr_t>(ptr) & 255u) == 0, "scratch symbol null or misaligned");
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return T5VQ_OK;
}
}  // namespace

int t5vq_vq_forward_noalloc_f32(const float* x, const float* w, const int64_t* range, const float* xx_in,
                                const float* ee_in, int64_t N, int64_t K, int64_t D, int loss_kind, float* out_ste,
                                float* out_q, float* out_diff, int64_t* idx_out, float* counts, float* embed_sum,
                                float* loss_terms, float* loss_sum, t5vq_stream_t stream_) {
    T5VQ_REQUIRE(N > 0 && K > 0, "N and K must be positive");
    T5VQ_REQUIRE(N < (1LL << 31) && K < (1LL << 31), "N or K too large");
    if (FwdWorkspace(N, K).total > static_cast<int64_t>(T5VQ_NOALLOC_SCRATCH_BYTES)) {
        char need[64];
        snprintf(need, sizeof(need), "%lld", static_cast<long long>(FwdWorkspace(N, K).total));
        return fail(T5VQ_ERR_INVALID,
                    "t5vq_vq_forward_noalloc_f32: N, K need %s bytes of temporaries, the static scratch holds "
                    "T5VQ_NOALLOC_SCRATCH_BYTES%s", need, " (use t5vq_vq_forward_f32 with a workspace)");
    }
    void* ptr = nullptr;
    const int rc = scratch_for_current_device(&ptr);
    if (rc != T5VQ_OK) return rc;
    // every temporary is (re)written before it is read: keys are filled with 'none', the statistics and the sort
    // cursors zeroed by the preparation tail of the first launch (or by the self-contained kernel's workgroups); nothing
    // of the previous call's contents is used
    return vq_forward_impl(x, w, range, xx_in, ee_in, N, K, D, loss_kind, ptr, out_ste, out_q, out_diff, idx_out,
                           counts, embed_sum, loss_terms, loss_sum, static_cast<hipStream_t>(stream_));
}

// The same forward with the filtered search (t5vq_filter.hip) in place of the scan: [fp16 images + row norms] ->
// [fp16 filter] -> [collect] -> [refine] -> [scan of the unsettled tokens] -> ([order tokens by code]) ->
// [gather + statistics] -> [loss sum].  Workspace = the scan form's, then the filter's.
int64_t t5vq_forward_filtered_workspace_bytes(int64_t N, int64_t K, int64_t D) {
    if (N < 0 || K < 0 || D < 0) return 0;
    return FwdWorkspace(N, K).total + t5vq_filter_workspace_bytes(N, K, D);
}

int t5vq_vq_forward_filtered_f32(const float* x, const float* w, const int64_t* range, const float* xx_in,
                                 const float* ee_in, int64_t N, int64_t K, int64_t D, int loss_kind, void* workspace,
                                 int64_t workspace_bytes, float* out_ste, float* out_q, float* out_diff,
                                 int64_t* idx_out, float* counts, float* embed_sum, float* loss_terms, float* loss_sum,
                                 int32_t* n_overflow, t5vq_stream_t stream_) {
    return t5vq_vq_forward_filtered_img_f32(x, w, range, xx_in, ee_in, N, K, D, loss_kind, workspace, workspace_bytes,
                                            out_ste, out_q, out_diff, idx_out, counts, embed_sum, loss_terms, loss_sum,
                                            n_overflow, nullptr, 0, nullptr, stream_);
}
int t5vq_vq_forward_filtered_img_f32(const float* x, const float* w, const int64_t* range, const float* xx_in,
                                     const float* ee_in, int64_t N, int64_t K, int64_t D, int loss_kind,
                                     void* workspace, int64_t workspace_bytes, float* out_ste, float* out_q,
                                     float* out_diff, int64_t* idx_out, float* counts, float* embed_sum,
                                     float* loss_terms, float* loss_sum, int32_t* n_overflow, void* code_image,
                                     int code_image_valid, const void* token_image, t5vq_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    T5VQ_REQUIRE(N > 0 && K > 0, "N and K must be positive");
    const FwdWorkspace ws(N, K);
    const int64_t filter_bytes = t5vq_filter_workspace_bytes(N, K, D);
    T5VQ_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "workspace null or not 256-byte aligned");
    T5VQ_REQUIRE(workspace_bytes >= ws.total + filter_bytes,
                 "workspace smaller than t5vq_forward_filtered_workspace_bytes(N, K, D)");
    T5VQ_REQUIRE(x && w, "null input");
    char* base = static_cast<char*>(workspace);
    int64_t* keys = reinterpret_cast<int64_t*>(base + ws.keys);
    // given norms are used as they are (they are inputs of the search then); the others are written by its first pass
    float* xx = xx_in ? const_cast<float*>(xx_in) : reinterpret_cast<float*>(base + ws.xx);
    float* ee = ee_in ? const_cast<float*>(ee_in) : reinterpret_cast<float*>(base + ws.ee);
    const int compute_norms = xx_in ? (ee_in ? 0 : 2) : (ee_in ? 3 : 1);
    float* partial = loss_terms ? loss_terms : (loss_sum ? reinterpret_cast<float*>(base + ws.partial) : nullptr);
    // index-only keys: the distance word of a token with a single candidate is an upper bound, not the distance
    // the search's first kernel also zeroes the statistics and the sort counters (three memset launches less)
    const bool order = embed_sum && N >= GS_SMALL_N;
    const bool fused_zero = (!counts || aligned16(counts)) && (!embed_sum || aligned16(embed_sum));
    T5vqZeroArgs zero = {};
    if (fused_zero) {
        zero.p[0] = counts;
        zero.n[0] = counts ? K : 0;
        zero.p[1] = embed_sum;
        zero.n[1] = embed_sum ? K * D : 0;
    }
    if (order) {
        zero.p[2] = base + ws.work;
        zero.n[2] = K + 2;
    }
    const int32_t* role_order = nullptr;
    if (range && N >= BUCKET_MIN_TOKENS) {   // as in the scan form: tokens visited role by role
        int32_t* p = reinterpret_cast<int32_t*>(base + ws.perm);
        const int rc0 = order_by_range(range, N, K, 0, p, reinterpret_cast<int32_t*>(base + ws.work), stream);
        if (rc0 != T5VQ_OK) return rc0;
        role_order = p;
    }
    int rc = t5vq_internal_dist_argmin_filtered(x, w, xx, ee, compute_norms, range, role_order, N, K, D, 0, 0,
                                                base + ws.total, filter_bytes, keys, n_overflow, zero, code_image,
                                                code_image_valid, token_image, 0, stream);
    if (rc != T5VQ_OK) return rc;
    const int32_t* perm = nullptr;
    if (order) {
        int32_t* p = reinterpret_cast<int32_t*>(base + ws.perm);
        rc = order_by_code(keys, N, K, p, reinterpret_cast<int32_t*>(base + ws.work), true, stream);
        if (rc != T5VQ_OK) return rc;
        perm = p;
    }
    rc = launch_gather(x, w, keys, N, K, D, loss_kind, out_ste, out_q, out_diff, idx_out, 0, K, counts, embed_sum,
                       partial, fused_zero, perm, stream);
    if (rc != T5VQ_OK || !loss_sum) return rc;
    return t5vq_reduce_partials_f32(partial, perm ? N : t5vq_gather_blocks(N), loss_sum, stream);
}

int t5vq_vq_ema_finalize_img_f32(float* cluster_size, float* ema_w, float* w, const float* counts,
                                 const float* embed_sum, const float* loss_sum, int64_t n_loss_terms, int64_t K,
                                 int64_t D, int64_t n_tokens, float loss_denom, float decay, float one_minus_decay,
                                 float epsilon, float k_epsilon, float commitment_cost, int plain_variant, int update,
                                 float* scalars, float* ee_out, void* code_image, t5vq_stream_t stream) {
    T5VQ_REQUIRE(!update || (cluster_size && ema_w && w && embed_sum), "update needs cluster_size, ema_w, w, embed_sum");
    int rc = t5vq_ema_cluster_f32(update ? cluster_size : nullptr, counts, loss_sum, n_loss_terms, K, n_tokens,
                                  loss_denom, decay, one_minus_decay, epsilon, k_epsilon, commitment_cost,
                                  plain_variant, update ? 1 : 0, scalars, stream);
    if (rc != T5VQ_OK || !update) return rc;
    return t5vq_ema_embed_img_f32(ema_w, w, embed_sum, cluster_size, K, D, decay, one_minus_decay, ee_out, code_image,
                                  stream);
}
int t5vq_vq_ema_finalize_f32(float* cluster_size, float* ema_w, float* w, const float* counts,
                             const float* embed_sum, const float* loss_sum, int64_t n_loss_terms, int64_t K, int64_t D,
                             int64_t n_tokens, float loss_denom, float decay, float one_minus_decay, float epsilon,
                             float k_epsilon, float commitment_cost, int plain_variant, int update, float* scalars,
                             float* ee_out, t5vq_stream_t stream) {
    return t5vq_vq_ema_finalize_img_f32(cluster_size, ema_w, w, counts, embed_sum, loss_sum, n_loss_terms, K, D, n_tokens,
                                        loss_denom, decay, one_minus_decay, epsilon, k_epsilon, commitment_cost,
                                        plain_variant, update, scalars, ee_out, nullptr, stream);
}

}  // extern "C"
