// t5vq_filter.hip -- exact nearest-code search by filtering in fp16 and deciding in fp32 (gfx950).
//
// The fused fp32 scan (t5vq_kernels.hip) is bound by the fp32 matrix pipe; the fp16 / bf16 pipe is 16x faster.  Here
// the fp16 pipe only FILTERS: for every token it leaves the few codes that could be the fp32 argmin, and the fp32
// arithmetic of the scan kernel is then repeated for those codes alone.  The result is the scan kernel's result,
// bit for bit, ties included.
//
//   d(i,k)   = fl(fl(xx_i + ee_k) - 2 dot(i,k)),  dot = the k-ordered fp32 fma chain of the scan kernel
//   d~(i,k)  = the same expression with dot~ = fp16(x_i) . fp16(e_k) (rows scaled by powers of two) accumulated in fp32 by the MFMA
//   |d~ - d| <= delta(i,k) = c2 |x_i| |e_k| + r (xx_i + ee_k)          (bound derived in DESIGN.md section 8)
//   T_i = min_k (d~ + delta)                     >= d(i, k*) for the true argmin k*
//   candidates {k : d~ - delta <= T_i}           contain k* and every code tied with it
//   refine: exact d for the candidates in the scan kernel's summation order, packed-key minimum
// One fp16 pass: every scanner (a fixed subset of a token's codes) keeps its smallest upper bound and its KEEP
// smallest lower bounds; collect_kernel forms T_i and filters the kept pairs.  A token with more than CAND_MAX
// candidates, or with a scanner that may have dropped one, is decided by the fp32 scan kernel run on those tokens only.
#include <hip/hip_runtime.h>
#include <atomic>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "t5vq_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr long long KEY_NONE = 0x7FFFFFFFFFFFFFFFLL;
constexpr int CAND_MAX = 16;            // candidates kept per token; more -> exact scan of all codes for that token
constexpr int KEEP = 4;                 // lowest lower bounds each scanner keeps (a scanner = one lane half of one
                                        // wave of one code split: it sees a fixed subset of the codes of its token)
#ifdef T5VQ_FILTER_W4   // experiment: 4 waves of 128 tokens x 128 codes (16 accumulators each) instead of 8 of 64 x 128
constexpr int TOK_TILE = 256, CODE_TILE = 256, THREADS = 256;
#else
constexpr int TOK_TILE = 256, CODE_TILE = 256, THREADS = 512;
#endif
constexpr int BKC = 32;                 // 16-byte-aligned chunk of a row, in 4-byte units = 64 fp16
constexpr int ROW = BKC;                // unpadded 128-byte rows (an LDS-DMA instruction writes 1 KB contiguously); the
                                        // 16-byte slot of piece p in row r is p ^ ((r >> 1) & 7): conflict-free b128 reads
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
constexpr int X_IMG = TOK_TILE * ROW, W_IMG = CODE_TILE * ROW, STAGE = X_IMG + W_IMG;
constexpr int EPI = 3 * CODE_TILE;         // per code of the tile: |e|^2, |e|, inverse row scale (epilogue operands)
constexpr int LDS_BYTES = (2 * STAGE + 2 * EPI) * 4;   // 131,072 + 6,144 B: one workgroup (8 waves) per CU
constexpr int F4_PER_ROW = BKC / 4, ROWS_PER_PASS = THREADS / F4_PER_ROW, XV = TOK_TILE / ROWS_PER_PASS,
              WV = CODE_TILE / ROWS_PER_PASS, G = BKC / 8;
#ifdef T5VQ_FILTER_W4
constexpr int TB = 4, CB = 4;           // 32x32 blocks per wave: 128 tokens x 128 codes
#else
constexpr int TB = 2, CB = 4;           // 32x32 blocks per wave: 64 tokens x 128 codes
#endif
constexpr int WAVES_T = TOK_TILE / (32 * TB);   // waves side by side along the tokens (the rest along the codes)

__device__ __forceinline__ int order_image(float d) {
    const int b = __float_as_int(d);
    return b >= 0 ? b : (b ^ 0x7FFFFFFF);
}
__device__ __forceinline__ int order_image_inv(int b) { return b >= 0 ? b : (b ^ 0x7FFFFFFF); }   // an involution
__device__ __forceinline__ long long pack_key(float d, int idx) {
    return (static_cast<long long>(order_image(d)) << 32) | static_cast<unsigned int>(idx);
}
// ---- preparation: fp16 images of x and w, every row scaled by a power of two so that its largest magnitude lies in
// [2^14, 2^15): the scaling is exact, rows of any magnitude (dead codes reach 1e6) fit fp16's range, and elements
// down to 2^-28 of the row maximum keep 11 significant bits.  inv[row] = the inverse factor (a power of two).
// One wave per row; se[k] = sqrt(ee[k]) for the error bound.
// norm_x / norm_w: this pass also writes |x|^2 / |e|^2 (otherwise they are inputs).
// order (nullable, ranged search): image row p is made from token order[p] -- the filter kernel then walks the tokens
// role by role although x itself is never moved; everything indexed by image row (xh, x_inv, tshare, the scanner
// records) is internal, everything the caller sees (xx, keys) stays indexed by token.
template <int NCH>
__global__ __launch_bounds__(256) void filter_prep_kernel(const float* __restrict__ x, long long n_x,
                                                          _Float16* __restrict__ xh, float* __restrict__ x_inv,
                                                          const float* __restrict__ w, long long n_w,
                                                          _Float16* __restrict__ wh, float* __restrict__ w_inv,
                                                          float* __restrict__ xx, float* __restrict__ ee,
                                                          long long kpad, float* __restrict__ se, int D,
                                                          int* __restrict__ n_list, int* __restrict__ tshare,
                                                          unsigned row_blocks, T5vqZeroArgs zero, int norm_x,
                                                          int norm_w, const int* __restrict__ order, int have_x_image,
                                                          const long long* __restrict__ range, long long code_lo,
                                                          long long code_hi) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x >= row_blocks) {   // the tail of the grid zeroes the caller's buffers: 4096 words per workgroup
        long long i = (static_cast<long long>(blockIdx.x - row_blocks) * 256 + threadIdx.x) * 16;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (i < zero.n[r]) {
#pragma unroll
                for (int j = 0; j < 16; j += 4) {
                    float* dst = static_cast<float*>(zero.p[r]) + i + j;
                    if (i + j + 4 <= zero.n[r]) {
                        *reinterpret_cast<f32x4*>(dst) = z;
                    } else {
                        for (int e = 0; e < 4; ++e)
                            if (i + j + e < zero.n[r]) dst[e] = 0.f;
                    }
                }
                return;
            }
            i -= (zero.n[r] + 4095) / 4096 * 4096;
            if (i < 0) return;
        }
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        n_list[0] = 0;   // tokens left to the exact scan
        n_list[1] = 0;   // (token, candidate) pairs left to refine_kernel
    }
    long long row = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const float* src;
    _Float16* dst;
    float* inv;
    float* norm;      // NORMS: where this row's squared norm goes (xx or ee)
    const bool is_w = row >= n_x;
    if (!is_w) {
        const long long token = order ? order[row] : row;
        src = x + token * D; dst = xh + row * D; inv = x_inv + row; norm = xx + token;
        if (lane == 0) tshare[row] = 0x7F800000;   // order image of +inf: no bound yet
        if (have_x_image) return;                  // the producer of x wrote this row's image (and |x|^2 is given)
        // ranged search on a codebook slice: a token whose range holds no code of the slice is never scored (its codes
        // are all masked), so its image row is not made -- one code shard of a grouped codebook sees one token in eight
        if (range && !norm_x) {
            const long long a = range[2 * token], e = range[2 * token + 1];
            if ((e < code_hi ? e : code_hi) <= (a > code_lo ? a : code_lo)) return;
        }
    } else {
        row -= n_x;
        if (row >= kpad) return;
        if (row >= n_w) {   // padding of the |e|^2 buffers: +inf, so padded codes can never be candidates
            if (lane == 0) {
                if (norm_w) ee[row] = __int_as_float(0x7F800000);
                se[row] = __int_as_float(0x7F800000);
                w_inv[row] = 1.0f;
            }
            return;
        }
        src = w + row * D; dst = wh + row * D; inv = w_inv + row; norm = ee + row;
    }
    f32x4 v[NCH];
    float m = 0.f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c4 = j * 64 + lane;
        v[j] = (c4 * 4 < D) ? reinterpret_cast<const f32x4*>(src)[c4] : zero4;
        m = t5vq_absmax4(m, v[j].x, v[j].y, v[j].z, v[j].w);
    }
    float sq = 0.f;
    if (is_w ? norm_w : norm_x) {   // the squared norm, in row_sqnorms_kernel's lane layout and order (bit-identical to it)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            sq = fmaf(v[j].x, v[j].x, sq);
            sq = fmaf(v[j].y, v[j].y, sq);
            sq = fmaf(v[j].z, v[j].z, sq);
            sq = fmaf(v[j].w, v[j].w, sq);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        if (lane == 0) *norm = sq;
    } else if (is_w) {
        sq = *norm;
    }
    if (is_w && lane == 0) se[row] = sqrtf(sq);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float invv;
    const float scale = t5vq_image_scale(m, &invv);
    if (lane == 0) *inv = invv;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c4 = j * 64 + lane;
        if (c4 * 4 < D) t5vq_image_store4(dst, c4, v[j].x, v[j].y, v[j].z, v[j].w, scale);
    }
}

// per-scanner state of one token: smallest upper bound seen, and the KEEP smallest lower bounds with their codes
struct Scan {
    float umin;
    float kl[KEEP];
    int kc[KEEP];
};
__device__ __forceinline__ void scan_init(Scan& s) {
    s.umin = __int_as_float(0x7F800000);
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        s.kl[i] = __int_as_float(0x7F800000);
        s.kc[i] = -1;
    }
}
__device__ __forceinline__ void scan_insert(Scan& s, float l, int c) {  // keeps kl ascending; no-op if l >= kl[KEEP-1]
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const bool lt = l < s.kl[i];
        const float tl = s.kl[i];
        const int tc = s.kc[i];
        s.kl[i] = lt ? l : tl;
        s.kc[i] = lt ? c : tc;
        l = lt ? tl : l;
        c = lt ? tc : c;
    }
}

// ---- the fp16 scan.  Workgroup = 8 waves, tile = 256 tokens x 256 codes, K-chunk = 64 fp16; wave (wt, wc) owns
// tokens [64 wt, +64) x codes [128 wc, +128): 2 x 4 accumulators of v_mfma_f32_32x32x16_f16.  Operand layout as in
// the fp32 scan kernel (token on the lane, codes in the accumulator registers); a 16-byte LDS fragment holds the
// 8 fp16 a lane feeds one MFMA (lane (r, h): k = 8h + j of a 16-deep step).  D2 = D / 2 = row length in 4-byte units.
// Output per scanner and token: umin and the KEEP kept (lower bound, code) pairs, token-minor for coalescing.
// RANGED (disentangled codebook): token i may only take codes [start_i, end_i) (global ids; this codebook slice
// starts at code_offset); codes outside get U = +inf and L = NaN, and code tiles outside the workgroup's ranges are
// skipped (the module buckets tokens by role, so a workgroup sees one or two ranges).
template <bool RANGED>
__global__ __launch_bounds__(THREADS, 1) void filter_f16_kernel(
    const float* __restrict__ xb, const float* __restrict__ wb, const float* __restrict__ xx,
    const float* __restrict__ ee, const float* __restrict__ se, const float* __restrict__ x_inv,
    const float* __restrict__ w_inv, int N, int K, int kpad, int D2, int n_splits, int tiles_per_split, int plain_map,
    float c2, float r, const long long* __restrict__ range, long long code_offset, float* __restrict__ ubuf,
    float* __restrict__ lbuf, int* __restrict__ cbuf, int* __restrict__ tshare, const int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int wt = wave % WAVES_T, wc = wave / WAVES_T;
    // Large problems: the splits of one token tile get ids 8 apart (one XCD, shared X tile in its L2).  Small problems
    // (plain_map: at most a few workgroups per CU in total): consecutive ids are different token tiles, so the
    // workgroups spread evenly over the XCDs -- with 12 token tiles the XCD-affine map loads 4 XCDs twice as much.
    int split, tok_base;
    if (plain_map) {
        const int tok_tiles = (N + TOK_TILE - 1) / TOK_TILE;
        split = blockIdx.x / tok_tiles;
        tok_base = (blockIdx.x % tok_tiles) * TOK_TILE;
    } else {
        const int grp = blockIdx.x / (8 * n_splits);
        split = (blockIdx.x / 8) % n_splits;
        tok_base = (grp * 8 + (blockIdx.x & 7)) * TOK_TILE;
    }
    if (tok_base >= N || split >= n_splits) return;
    const int n_code_tiles = (K + CODE_TILE - 1) / CODE_TILE;
    int tile_begin = split * tiles_per_split;
    int tile_end = tile_begin + tiles_per_split;
    if (tile_end > n_code_tiles) tile_end = n_code_tiles;
    const int n_chunks = D2 / BKC;  // D % 64 == 0 is required by the launcher

    Scan sc[TB];
    int tok[TB];
    int ls[TB], le[TB];   // RANGED: admissible local codes [ls, le)
    float xv[TB], csx[TB], m2[TB];
#pragma unroll
    for (int b = 0; b < TB; ++b) {
        scan_init(sc[b]);
        tok[b] = tok_base + wt * (32 * TB) + b * 32 + l31;
        const int tc = tok[b] < N ? tok[b] : N - 1;                    // image row
        const int real = (RANGED && order) ? order[tc] : tc;           // the token it was made from
        xv[b] = xx[real];
        csx[b] = c2 * sqrtf(xv[b]);
        m2[b] = -2.0f * x_inv[tc];   // -2 and the token row's inverse scale: exact (power of two)
        ls[b] = 0;
        le[b] = K;
        if (RANGED) {
            long long a = range[2 * static_cast<size_t>(real)] - code_offset;
            long long e = range[2 * static_cast<size_t>(real) + 1] - code_offset;
            a = a < 0 ? 0 : (a > K ? K : a);
            e = e < 0 ? 0 : (e > K ? K : e);
            ls[b] = static_cast<int>(a);
            le[b] = static_cast<int>(e);
        }
    }
    if (RANGED) {   // the workgroup's [lowest start, highest end): code tiles outside it are not visited
        int lo_c = 0x7FFFFFFF, hi_c = 0;
#pragma unroll
        for (int b = 0; b < TB; ++b)
            if (tok[b] < N && le[b] > ls[b]) {
                lo_c = min(lo_c, ls[b]);
                hi_c = max(hi_c, le[b]);
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo_c = min(lo_c, __shfl_xor(lo_c, o));
            hi_c = max(hi_c, __shfl_xor(hi_c, o));
        }
        int* red = reinterpret_cast<int*>(lds);
        if (lane == 0) {
            red[2 * wave] = lo_c;
            red[2 * wave + 1] = hi_c;
        }
        __syncthreads();
        for (int i = 0; i < THREADS / 64; ++i) {
            lo_c = min(lo_c, red[2 * i]);
            hi_c = max(hi_c, red[2 * i + 1]);
        }
        __syncthreads();   // the staging writes below reuse this LDS
        tile_begin = max(tile_begin, lo_c / CODE_TILE);
        tile_end = min(tile_end, (hi_c + CODE_TILE - 1) / CODE_TILE);
    }
    const int n_tiles = tile_end - tile_begin;
    const int total = n_tiles * n_chunks;
    // RANGED: none of this split's codes lies in the ranges of this workgroup's tokens -- collect_kernel does not read
    // the records of a (token, split) whose range misses the split, so nothing is written (workgroup-uniform)
    if (RANGED && n_tiles <= 0) return;

    if (n_tiles > 0) {
        // LDS-DMA staging (global_load_lds_dwordx4): lane L of a wave-instruction fills the 16 bytes at base + 16 L =
        // row (L / 8), slot (L % 8) of an 8-row slab; slot s of row r holds piece s ^ ((r >> 1) & 7) of the row's 8
        // 16-byte pieces -- the swizzle sits on the SOURCE address, the LDS image stays lane-linear.  No VGPR round
        // trip and no ds_write pass (the register form needed 32 staging VGPRs and spilled in the ranged kernel).
        const int s_row = tid / F4_PER_ROW;                                   // row inside a 64-row pass
        const int s_col = ((tid % F4_PER_ROW) ^ ((s_row >> 1) & 7)) * 4;      // source piece (4-byte units)
        const float* xrow[XV];
#pragma unroll
        for (int j = 0; j < XV; ++j) {
            int t = tok_base + s_row + ROWS_PER_PASS * j;
            if (t > N - 1) t = N - 1;
            xrow[j] = xb + static_cast<size_t>(t) * D2 + s_col;
        }
        auto issue_stage = [&](int buf, int ct, int kc) {
            float* xs = lds + buf * STAGE + wave * 8 * ROW;
            float* ws = xs + X_IMG;
#pragma unroll
            for (int j = 0; j < XV; ++j)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(xrow[j] + kc * BKC),
                                                 (lds_ptr_t)(xs + ROWS_PER_PASS * j * ROW), 16, 0, 0);
#pragma unroll
            for (int j = 0; j < WV; ++j) {
                int c = (tile_begin + ct) * CODE_TILE + s_row + ROWS_PER_PASS * j;
                if (c > K - 1) c = K - 1;
                __builtin_amdgcn_global_load_lds(
                    (gbl_ptr_t)(wb + static_cast<size_t>(c) * D2 + s_col + kc * BKC),
                    (lds_ptr_t)(ws + ROWS_PER_PASS * j * ROW), 16, 0, 0);
            }
        };
        // Two buffers: stage it+1 is issued at the top of step it (its buffer is free since the barrier that ended
        // step it-1) and has to land by the wait + barrier that end step it -- the DMA's data is ordered for a
        // ds_read only by the issuing wave's vmcnt wait followed by a barrier the reader has passed.
        issue_stage(0, 0, 0);
        int lct = 0, lkc = 0;                       // (code tile, chunk) of the stage issued last
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int rd_sw = (l31 >> 1) & 7;           // the reader's rows are 32 k + l31: (row >> 1) & 7 is the lane's
        int it = 0;
        for (int ct = 0; ct < n_tiles; ++ct) {
            f32x16 acc[TB][CB];
#pragma unroll
            for (int b = 0; b < TB; ++b)
#pragma unroll
                for (int t = 0; t < CB; ++t)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[b][t][q] = 0.f;
            // the epilogue's per-code operands go through LDS (two images, by tile parity): fetched during the first
            // chunk of the tile, written during the second, read after the tile's last barrier -- read from global
            // memory in the epilogue itself, every group of four codes waited a full L2 latency
            float* epi = lds + 2 * STAGE + (ct & 1) * EPI;
            float ep0 = 0.f, ep1 = 0.f, ep2 = 1.f;
            int shared_u[TB];
            for (int kc = 0; kc < n_chunks; ++kc, ++it) {
                if (kc == 0) {
                    const int c = (tile_begin + ct) * CODE_TILE + (tid & (CODE_TILE - 1));
                    if (tid < CODE_TILE) {
                        ep0 = c < kpad ? ee[c] : __int_as_float(0x7F800000);
                        ep1 = c < kpad ? se[c] : 0.f;
                        if (THREADS < 2 * CODE_TILE) ep2 = c < kpad ? w_inv[c] : 1.f;
                    } else {
                        ep0 = c < kpad ? w_inv[c] : 1.f;
                    }
                }
                if (kc == n_chunks - 1) {   // the bound other scanners of these tokens have published (any split, any CU)
#pragma unroll
                    for (int b = 0; b < TB; ++b)   // device-scope load: a plain one may hit this CU's stale L1 line
                        shared_u[b] = __hip_atomic_load(tshare + (tok[b] < N ? tok[b] : N - 1), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
                }
                if (kc == (n_chunks > 1 ? 1 : 0)) {
                    if (tid < CODE_TILE) {
                        epi[tid] = ep0;
                        epi[CODE_TILE + tid] = ep1;
                        if (THREADS < 2 * CODE_TILE) epi[2 * CODE_TILE + tid] = ep2;
                    } else {
                        epi[2 * CODE_TILE + (tid & (CODE_TILE - 1))] = ep0;
                    }
                }
                const float* xs = lds + (it & 1) * STAGE + (wt * (32 * TB) + l31) * ROW;
                const float* ws = lds + (it & 1) * STAGE + X_IMG + (wc * 128 + l31) * ROW;
                // stage it+1 goes straight into the other buffer (free since the barrier that ended step it-1) and
                // must have landed by this step's closing barrier; past the end the last stage is fetched again
                if (it + 1 < total && ++lkc == n_chunks) { lkc = 0; ++lct; }
                issue_stage((it + 1) & 1, lct, lkc);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    f16x8 bf[TB], af[CB];
                    const int piece = ((2 * g + h) ^ rd_sw) * 4;
#pragma unroll
                    for (int b = 0; b < TB; ++b)
                        bf[b] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(xs + b * 32 * ROW + piece));
#pragma unroll
                    for (int t = 0; t < CB; ++t)
                        af[t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(ws + t * 32 * ROW + piece));
#pragma unroll
                    for (int t = 0; t < CB; ++t)
#pragma unroll
                        for (int b = 0; b < TB; ++b)
                            acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], bf[b], acc[b][t], 0, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            // epilogue: d~ = fl(fl(xx + ee) - 2 dot~), delta = c2 |x| |e| + r (xx + ee); U = d~ + delta, L = d~ - delta.
            // Fast path, per (token, 32-code block): only d~ and its minimum are formed per element; the bounds use
            // delta_max = delta at the largest |e| and |e|^2 of the block (rounding is monotone, so delta <= delta_max
            // for each of its codes): umin <- min(umin, min d~ + delta_max) stays an upper bound of the true minimum,
            // and no code of a group of four can be a candidate unless min4 d~ - delta_max <= umin.  Groups that pass
            // (a few per cent) take the per-element path with each code's own delta, so the kept lower bounds are as
            // tight as before.
#ifdef T5VQ_AB_NOEPI
            {
                float keep = 0.f;
#pragma unroll
                for (int b = 0; b < TB; ++b)
#pragma unroll
                    for (int t = 0; t < CB; ++t)
#pragma unroll
                        for (int q = 0; q < 16; ++q) keep += acc[b][t][q];
                if (keep == 12345.678f) sc[0].umin = keep;
                continue;
            }
#endif
            // Scanners of one token share their smallest upper bound through memory (order image, atomic min): each
            // scanner alone would let every running minimum of ITS codes through as a candidate; with the shared
            // bound only codes near the minimum over all codes seen so far by anyone pass.  Any published value is
            // a valid upper bound of the token's true minimum, however stale.
#pragma unroll
            for (int b = 0; b < TB; ++b) sc[b].umin = fminf(sc[b].umin, __int_as_float(order_image_inv(shared_u[b])));
            const int blk0 = (tile_begin + ct) * CODE_TILE + wc * 128;
            const int code0 = blk0 + 4 * h;
#pragma unroll
            for (int t = 0; t < CB; ++t) {
                if (blk0 + t * 32 >= kpad) break;   // |e|^2 is padded to 128 codes, the tile spans 256
                const float* ep_e = epi + wc * 128 + t * 32;
                const float* ep_s = ep_e + CODE_TILE;
                const float* ep_i = ep_e + 2 * CODE_TILE;
                float egmax = ep_e[l31], sgmax = ep_s[l31];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {
                    egmax = fmaxf(egmax, __shfl_xor(egmax, o));   // padded codes: +inf -> delta_max = +inf -> slow path
                    sgmax = fmaxf(sgmax, __shfl_xor(sgmax, o));
                }
                float dmax[TB];
#pragma unroll
                for (int b = 0; b < TB; ++b) dmax[b] = fmaf(csx[b], sgmax, r * (xv[b] + egmax));
                const bool padded = !(egmax < __int_as_float(0x7F800000));   // a block with padded codes is looked at
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int cb = code0 + t * 32 + 8 * g4;
                    const f32x4 e4 = *reinterpret_cast<const f32x4*>(ep_e + 4 * h + 8 * g4);
                    const f32x4 i4 = *reinterpret_cast<const f32x4*>(ep_i + 4 * h + 8 * g4);
#pragma unroll
                    for (int b = 0; b < TB; ++b) {
                        float dm = __int_as_float(0x7F800000);
                        float sv[4], dv[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            sv[c] = xv[b] + e4[c];
                            dv[c] = fmaf(acc[b][t][g4 * 4 + c] * m2[b], i4[c], sv[c]);   // -2 dot~, rescaled exactly
                            float d = dv[c];
                            if (RANGED) d = (cb + c >= ls[b] && cb + c < le[b]) ? d : __int_as_float(0x7F800000);
                            dm = fminf(dm, d);
                        }
                        sc[b].umin = fminf(sc[b].umin, dm + dmax[b]);   // inf + inf = inf; never NaN (d~ > -inf)
                        const float lm = dm - dmax[b];                  // inf - inf = NaN compares false: `padded`
#ifdef T5VQ_AB_NOSLOW
                        if (lm != 12345.678f) continue;
#endif
                        if (!__any(padded || (lm <= sc[b].umin && lm < sc[b].kl[KEEP - 1]))) continue;
                        const f32x4 s4 = *reinterpret_cast<const f32x4*>(ep_s + 4 * h + 8 * g4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float d = dv[c];
                            const float delta = fmaf(csx[b], s4[c], r * sv[c]);
                            float up = d + delta;                        // padded codes: +inf
                            float lo = d - delta;                        // padded codes: inf - inf = NaN
                            if (RANGED) {
                                const bool ok = cb + c >= ls[b] && cb + c < le[b];
                                up = ok ? up : __int_as_float(0x7F800000);
                                lo = ok ? lo : __int_as_float(0x7FC00000);   // NaN: never a candidate
                            }
                            sc[b].umin = fminf(sc[b].umin, up);          // the code's own, tighter bound
                            if (lo <= sc[b].umin) scan_insert(sc[b], lo, cb + c);   // NaN compares false
                        }
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                sc[b].umin = fminf(sc[b].umin, __shfl_xor(sc[b].umin, 32));   // the two lane halves scan one token
                if (h == 0 && tok[b] < N) atomicMin(tshare + tok[b], order_image(sc[b].umin));
            }
        }
    }
    const int scanner = (split * 2 + wc) * 2 + h;
#pragma unroll
    for (int b = 0; b < TB; ++b) {
        if (tok[b] < N) {
            ubuf[static_cast<size_t>(scanner) * N + tok[b]] = sc[b].umin;
#pragma unroll
            for (int i = 0; i < KEEP; ++i) {
                lbuf[(static_cast<size_t>(scanner) * KEEP + i) * N + tok[b]] = sc[b].kl[i];
                cbuf[(static_cast<size_t>(scanner) * KEEP + i) * N + tok[b]] = sc[b].kc[i];
            }
        }
    }
}

// ---- collect: T_i = min over scanners of umin; candidates = kept pairs with lower bound <= T_i.  A scanner whose
// KEEP kept pairs all qualify may have dropped further ones: the token is then decided by the exact scan.
// L lanes per token, one per scanner (L = 4, 8, 16, 32 >= n_scanners): with one thread per token a small batch with
// many code splits walked 32 x 9 dependent loads in 12 workgroups (C1: 76 us).  Candidates keep the (scanner, slot)
// order through a prefix sum over the token's lanes (any order would do: refine takes the packed-key minimum).
constexpr int COLLECT_THREADS = 1024;
// Every thread of the workgroup asks for `mine` consecutive slots of a list whose length lives in *counter: one
// atomic per workgroup (additions to one word from all XCDs serialise at about 10 ns each: one per wave cost 32 us
// at C3).  Returns the thread's first slot.  All threads of the workgroup must call it.
__device__ __forceinline__ int block_reserve(int mine, int* counter, int* s_wave_total, int* s_block_base) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    __syncthreads();   // the previous use of the two shared words is over
    if (lane == 63) s_wave_total[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int i = 0; i < COLLECT_THREADS / 64; ++i) total += s_wave_total[i];
        *s_block_base = total > 0 ? atomicAdd(counter, total) : 0;
    }
    __syncthreads();
    int base = *s_block_base;
    for (int i = 0; i < wave; ++i) base += s_wave_total[i];
    return base + incl - mine;
}
template <int L, int R>   // R scanners per lane (2 only with L = 64: up to 128 scanners)
__global__ __launch_bounds__(COLLECT_THREADS) void collect_kernel(const float* __restrict__ ubuf, const float* __restrict__ lbuf,
                                                      const int* __restrict__ cbuf, int N, int K, int n_scanners,
                                                      int exact_keys, const long long* __restrict__ range,
                                                      long long code_offset, int codes_per_split,
                                                      int* __restrict__ pair_tok,
                                                      int* __restrict__ pair_code, long long* __restrict__ keys,
                                                      int* __restrict__ tok_list, int* __restrict__ n_list,
                                                      const int* __restrict__ order) {
    const long long gid = static_cast<long long>(blockIdx.x) * COLLECT_THREADS + threadIdx.x;
    const int tok = static_cast<int>(gid / L), s = static_cast<int>(gid % L);   // tok: image row (scanner records)
    // No lane leaves early: the pair counter below is reserved once per WORKGROUP (prefix sums over all its lanes).
    bool active = tok < N;
    const int real = (active && order) ? order[tok] : tok;   // the token: ranges, keys and the two lists are by token
    int ra = 0, re = K;      // the token's admissible local codes
    if (active && range) {   // a range that misses this codebook slice (other shard / empty): 'none', as the scan leaves it
        const long long a = range[2 * static_cast<size_t>(real)] - code_offset;
        const long long e = range[2 * static_cast<size_t>(real) + 1] - code_offset;
        ra = static_cast<int>(a > 0 ? (a < K ? a : K) : 0);
        re = static_cast<int>(e < K ? (e > 0 ? e : 0) : K);
        if (re <= ra) {
            if (s == 0) keys[real] = KEY_NONE;
            active = false;
        }
    }
    // a workgroup whose tokens all miss the slice is done (role order puts them side by side: most workgroups of a
    // search on a code shard); otherwise no lane leaves early, see above
    if (range && !__syncthreads_or(active ? 1 : 0)) return;
    const int n_live = active ? n_scanners : 0;   // an inactive token reads nothing and ends up with no candidates
    // ranged: the scanners of a code split the token's range misses hold no record (their workgroup may have left
    // without writing any) -- four scanners per split
    auto live = [&](int sc) {
        if (sc >= n_live) return false;
        if (!range) return true;
        const int lo = (sc >> 2) * codes_per_split;
        return max(ra, lo) < min(re, lo + codes_per_split);
    };
    float T = __int_as_float(0x7F800000);
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (live(s + r * L)) T = fminf(T, ubuf[static_cast<size_t>(s + r * L) * N + tok]);
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) T = fminf(T, __shfl_xor(T, o));
    int c[R * KEEP];
    bool ok[R * KEEP];
    int q = 0, first = -1, overflow = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int sc = s + r * L;
        int qs = 0;
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            float l = 0.f;
            int ci = -1;
            if (live(sc)) {
                l = lbuf[(static_cast<size_t>(sc) * KEEP + i) * N + tok];
                ci = cbuf[(static_cast<size_t>(sc) * KEEP + i) * N + tok];
            }
            c[r * KEEP + i] = ci;
            ok[r * KEEP + i] = ci >= 0 && l <= T;
            if (ok[r * KEEP + i] && first < 0) first = ci;
            qs += ok[r * KEEP + i] ? 1 : 0;
        }
        overflow |= qs == KEEP ? 1 : 0;
        q += qs;
    }
    int incl = q;   // inclusive prefix over the token's lanes
#pragma unroll
    for (int o = 1; o < L; o <<= 1) {
        const int v = __shfl_up(incl, o, L);
        if (s >= o) incl += v;
    }
    const int n = __shfl(incl, L - 1, L);
    int f = (q > 0 && incl == q) ? first : -1;   // the first qualifying pair of the first lane that has one
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) {
        overflow |= __shfl_xor(overflow, o);
        f = max(f, __shfl_xor(f, o));
    }
    // the same decision in every lane of the token
    const bool to_scan = overflow || n > CAND_MAX || n == 0;   // n == 0: NaN / inf rows -- whatever the scan does
    const bool decided = !to_scan && n == 1 && !exact_keys;
    // (token, candidate) pairs go to refine_kernel as one dense list, unsettled tokens to the scan's list
    __shared__ int s_wave_total[COLLECT_THREADS / 64], s_block_base;
    const bool lead = active && s == 0;
    const int pair_slot = block_reserve(lead && !to_scan && !decided ? n : 0, n_list + 1, s_wave_total, &s_block_base);
    const int list_slot = block_reserve(lead && to_scan ? 1 : 0, n_list, s_wave_total, &s_block_base);
    const int tok_base = __shfl(pair_slot, 0, L);   // the token's first pair slot, from its lane 0
    if (!active) return;
    if (to_scan) {
        if (s == 0) {
            keys[real] = KEY_NONE;              // the listed scan merges into it with atomicMin
            tok_list[list_slot] = real;
        }
    } else if (decided) {                       // the index is decided; T_i (an upper bound of its distance) fills
        if (s == 0) {
            const long long gi = code_offset + f - (range ? range[2 * static_cast<size_t>(real)] : 0);
            keys[real] = pack_key(T, static_cast<int>(gi));  // the key's distance word (index: slice-relative if ranged)
        }
    } else {   // refine_kernel merges the pairs' exact keys into the token's key with atomicMin
        if (s == 0) keys[real] = KEY_NONE;
        int slot = tok_base + incl - q;
#pragma unroll
        for (int i = 0; i < R * KEEP; ++i)
            if (ok[i]) {
                pair_tok[slot] = real;
                pair_code[slot] = c[i];
                ++slot;
            }
    }
}

// exact fp32 distance of one (token, code) pair in the scan kernel's arithmetic: k-ordered fma chain, within each
// group of 8 the order 0,4,1,5,2,6,3,7 (what its MFMA operand layout feeds the matrix pipe), then fl(fl(xx+ee) - 2 dot)
__device__ __forceinline__ float exact_distance(const float* __restrict__ xr, const float* __restrict__ wr, int D,
                                                float xxv, float eev) {
    float dot = 0.f;
    // D % 64 == 0; the 32 loads of an iteration (64 elements of both rows) are issued before its fma chain: a lane's rows
    // are its own (uncoalesced 16-byte pieces), the chain is latency-bound and the loads are what can overlap
    for (int k = 0; k < D; k += 64) {
        f32x4 xq[16], wq[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            xq[i] = *reinterpret_cast<const f32x4*>(xr + k + 4 * i);
            wq[i] = *reinterpret_cast<const f32x4*>(wr + k + 4 * i);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 x0 = xq[2 * g], x1 = xq[2 * g + 1], w0 = wq[2 * g], w1 = wq[2 * g + 1];
            dot = fmaf(w0.x, x0.x, dot);
            dot = fmaf(w1.x, x1.x, dot);
            dot = fmaf(w0.y, x0.y, dot);
            dot = fmaf(w1.y, x1.y, dot);
            dot = fmaf(w0.z, x0.z, dot);
            dot = fmaf(w1.z, x1.z, dot);
            dot = fmaf(w0.w, x0.w, dot);
            dot = fmaf(w1.w, x1.w, dot);
        }
    }
    return fmaf(-2.0f, dot, xxv + eev);
}

// ---- refine: one lane per (token, candidate) pair of the dense list collect_kernel wrote; the exact distances merge
// into the token's key with atomicMin (lowest distance, then lowest index: the scan's rule).  With one lane per
// (token, slot) -- 16 slots per token, 2 or 3 in use -- a batch where most tokens keep two candidates ran 16384 waves
// of 768-long fma chains at 15 % lane use (cold / collapsed regimes: 84 us); the host does not know the pair count,
// so a fixed grid strides over the list.
__global__ __launch_bounds__(256) void refine_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ xx, const float* __restrict__ ee,
                                                     int D, const long long* __restrict__ range,
                                                     long long code_offset, const int* __restrict__ counters,
                                                     const int* __restrict__ pair_tok,
                                                     const int* __restrict__ pair_code, long long* __restrict__ keys,
                                                     int* __restrict__ n_overflow) {
    // the caller's copy of the unsettled-token count (collect_kernel is complete: saves a copy launch)
    if (n_overflow && blockIdx.x == 0 && threadIdx.x == 0) *n_overflow = counters[0];
    // about 16 pairs per wave while the list is short (a dense wave touches 64 rows per load: few waves would carry
    // the whole list on a few CUs), all 64 lanes once every wave of the grid has that many
    const int n = counters[1];   // counters: [0] tokens left to the scan, [1] pairs
    const int n_waves = gridDim.x * 4;
    int w_eff = (n + 15) / 16;
    w_eff = w_eff < 1 ? 1 : (w_eff > n_waves ? n_waves : w_eff);
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave >= w_eff) return;
    for (long long p = static_cast<long long>(lane) * w_eff + wave; p < n; p += 64LL * w_eff) {
        const int tok = pair_tok[p], code = pair_code[p];
        const long long gi = code_offset + code - (range ? range[2 * static_cast<size_t>(tok)] : 0);
        const long long key = pack_key(exact_distance(x + static_cast<size_t>(tok) * D, w + static_cast<size_t>(code) * D,
                                                      D, xx[tok], ee[code]), static_cast<int>(gi));
        atomicMin(keys + tok, key);
    }
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
struct FilterWorkspace {
    int64_t xb, wb, se, xinv, winv, ubuf, lbuf, cbuf, cnt, cand, list, nlist, tshare, total;
    FilterWorkspace(int64_t N, int64_t K, int64_t D, int64_t scanners) {
        xb = 0;
        wb = align_up(xb + 2 * N * D, 256);
        se = align_up(wb + 2 * K * D, 256);
        xinv = align_up(se + 4 * t5vq_padded_codes(K), 256);
        winv = align_up(xinv + 4 * N, 256);
        ubuf = align_up(winv + 4 * t5vq_padded_codes(K), 256);
        lbuf = align_up(ubuf + 4 * N * scanners, 256);
        cbuf = align_up(lbuf + 4 * N * scanners * KEEP, 256);
        cnt = align_up(cbuf + 4 * N * scanners * KEEP, 256);       // pair list: token ids  [N * CAND_MAX]
        cand = align_up(cnt + 4 * N * CAND_MAX, 256);              // pair list: codes      [N * CAND_MAX]
        list = align_up(cand + 4 * N * CAND_MAX, 256);
        nlist = align_up(list + 4 * N, 256);
        tshare = align_up(nlist + 64, 256);                 // int32 [N]: smallest upper bound any scanner has seen
        total = align_up(tshare + 4 * N, 256);
    }
};
constexpr int MAX_SPLITS = 32;   // 4 scanners each: collect_kernel handles up to 128
constexpr int MIN_TILES = 1;

// code splits: the fewest code tiles per workgroup for which all workgroups still run at once (<= 256), at most
// MAX_SPLITS.  C1 (12 token tiles, 40 code tiles): 20 splits of 2 tiles, search 113 us; 8 splits of 5 tiles 165 us.
// (An earlier measurement had more splits slower: that was the one-thread-per-token collect kernel, not the splits.)
// Ranged search: a workgroup only visits the code tiles of its tokens' ranges, so how many workgroups do work is not
// known on the host (one code shard of a disentangled codebook: the tokens of one role in eight, and with one split
// 32 of 256 workgroups walked all 32 code tiles of the shard while the others had none: 869 us).  The codes are cut
// into at least RANGED_SPLITS fixed pieces whatever N is; workgroups whose piece misses their tokens' ranges leave
// after reading the ranges (no scanner records: collect_kernel skips the splits a token's range misses), and the
// dispatcher hands the others out as compute units free up.  4 pieces: that shard 328 us, whole codebooks of 8192 ..
// 65536 codes level or up to 1.35x faster; 8 is better on the aligned shard (266 us) and worse elsewhere
// (tools/experiments/ranged_filter_splits.py, profiles/r3_ranged_filter_splits.txt) -- the host does not see the
// ranges, so a caller who knows that most tokens belong elsewhere (a code shard) says so: sparse -> 8 pieces.
constexpr int64_t RANGED_SPLITS = 4, RANGED_SPLITS_SPARSE = 8;
int64_t plan_filter_splits(int64_t N, int64_t K, bool ranged, bool sparse = false) {
    const int64_t tok_tiles = (N + TOK_TILE - 1) / TOK_TILE;
    const int64_t code_tiles = (K + CODE_TILE - 1) / CODE_TILE;
    if (ranged) {
        int64_t s = sparse ? RANGED_SPLITS_SPARSE : RANGED_SPLITS;
#ifdef T5VQ_TUNING
        if (const char* e = getenv("T5VQ_RANGED_SPLITS")) s = atoi(e);
#endif
        const int64_t plain = plan_filter_splits(N, K, false);
        if (s <= plain) return plain;
        s = s < code_tiles ? s : code_tiles;
        const int64_t t = (code_tiles + s - 1) / s;
        return (code_tiles + t - 1) / t;               // no empty splits
    }
    // the fewest code tiles per workgroup (at least MIN_TILES) for which all workgroups still run at once
    for (int64_t t = MIN_TILES; t < code_tiles; ++t) {
        const int64_t splits = (code_tiles + t - 1) / t;
        if (splits <= MAX_SPLITS && tok_tiles * splits <= 256) return splits;
    }
    return 1;
}
std::atomic<bool> attr_done[64];   // idempotent set-up, may race between host threads

}  // namespace

extern "C" {

int64_t t5vq_filter_workspace_bytes(int64_t N, int64_t K, int64_t D) {
    if (N < 0 || K < 0 || D < 0) return 0;
    const int64_t a = plan_filter_splits(N, K, false), b = plan_filter_splits(N, K, true, true);   // the call may be any
    return FilterWorkspace(N, K, D, 4 * (a > b ? a : b)).total;
}

int t5vq_dist_argmin_filtered_f32(const float* x, const float* w, float* xx, float* ee, int compute_norms,
                                  const int64_t* range, const int32_t* order, int64_t N, int64_t K, int64_t D,
                                  int64_t code_offset, int exact_keys, void* workspace, int64_t workspace_bytes,
                                  int64_t* keys, int32_t* n_overflow, t5vq_stream_t stream_) {
    T5vqZeroArgs zero = {};
    return t5vq_internal_dist_argmin_filtered(x, w, xx, ee, compute_norms, range, order, N, K, D, code_offset, exact_keys,
                                              workspace, workspace_bytes, keys, n_overflow, zero, nullptr, 0, nullptr, 0,
                                              static_cast<hipStream_t>(stream_));
}

int64_t t5vq_code_image_bytes(int64_t K, int64_t D) {
    if (K <= 0 || D <= 0) return 0;
    return t5vq_align256(2 * K * D) + 2 * t5vq_align256(4 * t5vq_padded_codes(K));
}
int64_t t5vq_token_image_bytes(int64_t N, int64_t D) {
    if (N <= 0 || D <= 0) return 0;
    return t5vq_align256(2 * N * D) + t5vq_align256(4 * N);
}
int t5vq_dist_argmin_filtered_img_f32(const float* x, const float* w, float* xx, float* ee, int compute_norms,
                                      const int64_t* range, const int32_t* order, int64_t N, int64_t K, int64_t D,
                                      int64_t code_offset, int exact_keys, void* workspace, int64_t workspace_bytes,
                                      int64_t* keys, int32_t* n_overflow, void* code_image, int code_image_valid,
                                      const void* token_image, int range_sparse, t5vq_stream_t stream_) {
    T5vqZeroArgs zero = {};
    return t5vq_internal_dist_argmin_filtered(x, w, xx, ee, compute_norms, range, order, N, K, D, code_offset, exact_keys,
                                              workspace, workspace_bytes, keys, n_overflow, zero, code_image,
                                              code_image_valid, token_image, range_sparse,
                                              static_cast<hipStream_t>(stream_));
}

int t5vq_internal_dist_argmin_filtered(const float* x, const float* w, float* xx, float* ee, int compute_norms,
                                       const int64_t* range, const int32_t* order, int64_t N, int64_t K, int64_t D,
                                       int64_t code_offset, int exact_keys, void* workspace, int64_t workspace_bytes,
                                       int64_t* keys, int32_t* n_overflow, T5vqZeroArgs zero, void* code_image,
                                       int code_image_valid, const void* token_image, int range_sparse,
                                       hipStream_t stream) {
    T5VQ_I_REQUIRE(x && w && xx && ee && keys && workspace, "null pointer");
    if (order) token_image = nullptr;   // the search walks image rows in role order: an image in token order is no use
    const bool have_w_image = code_image && code_image_valid;
    const bool have_x_image = token_image != nullptr;
    T5VQ_I_REQUIRE(!have_w_image || compute_norms == 0 || compute_norms == 3,
                   "a valid code image comes with the codebook's norms: ee must be an input");
    T5VQ_I_REQUIRE(!have_x_image || compute_norms == 0 || compute_norms == 2,
                   "a token image comes with the row norms of x: xx must be an input");
    T5VQ_I_REQUIRE((reinterpret_cast<uintptr_t>(code_image) & 255u) == 0 && (reinterpret_cast<uintptr_t>(token_image) & 255u) == 0,
                   "image blocks must be 256-byte aligned");
    T5VQ_I_REQUIRE(!order || range, "order is an ordering for the ranged search: range is null");
    T5VQ_I_REQUIRE(compute_norms >= 0 && compute_norms <= 3, "compute_norms must be 0..3");
    T5VQ_I_REQUIRE(N > 0 && K > 0 && N < (1LL << 31) - TOK_TILE && K < (1LL << 31) - 512, "N, K out of range");
    T5VQ_I_REQUIRE(D >= 64 && D % 64 == 0 && D <= 4096, "the filtered search needs D % 64 == 0, 64 <= D <= 4096");
    T5VQ_I_REQUIRE(code_offset >= 0 && code_offset + K < (1LL << 31), "code_offset out of range");
    T5VQ_I_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (reinterpret_cast<uintptr_t>(w) & 15u) == 0 &&
                       (reinterpret_cast<uintptr_t>(ee) & 15u) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0,
                   "x, w, ee must be 16-byte aligned and the workspace 256-byte aligned");
    const int64_t splits = plan_filter_splits(N, K, range != nullptr, range_sparse != 0);
    const int64_t scanners = 4 * splits;
    const FilterWorkspace ws(N, K, D, scanners);
    T5VQ_I_REQUIRE(workspace_bytes >= ws.total, "workspace smaller than t5vq_filter_workspace_bytes(N, K, D)");
    int dev = 0;
    T5VQ_I_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_done[dev].load(std::memory_order_acquire)) {
        T5VQ_I_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&filter_f16_kernel<false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        T5VQ_I_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&filter_f16_kernel<true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done[dev].store(true, std::memory_order_release);
    }
    char* base = static_cast<char*>(workspace);
    _Float16* xb = reinterpret_cast<_Float16*>(base + ws.xb);
    _Float16* wb = reinterpret_cast<_Float16*>(base + ws.wb);
    float* se = reinterpret_cast<float*>(base + ws.se);
    float* x_inv = reinterpret_cast<float*>(base + ws.xinv);
    float* w_inv = reinterpret_cast<float*>(base + ws.winv);
    if (code_image) {    // the codebook's image lives in the caller's block (read there, or written there by this call)
        const T5vqCodeImage im = t5vq_code_image_parts(code_image, K, D);
        wb = im.rows, w_inv = im.inv, se = im.se;
    }
    if (have_x_image) {
        const T5vqTokenImage im = t5vq_token_image_parts(const_cast<void*>(token_image), N, D);
        xb = im.rows, x_inv = im.inv;
    }
    float* ubuf = reinterpret_cast<float*>(base + ws.ubuf);
    float* lbuf = reinterpret_cast<float*>(base + ws.lbuf);
    int* cbuf = reinterpret_cast<int*>(base + ws.cbuf);
    int* cnt = reinterpret_cast<int*>(base + ws.cnt);
    int* cand = reinterpret_cast<int*>(base + ws.cand);
    int* tok_list = reinterpret_cast<int*>(base + ws.list);
    int* n_list = reinterpret_cast<int*>(base + ws.nlist);
    int* tshare = reinterpret_cast<int*>(base + ws.tshare);
    const int64_t kpad = t5vq_padded_codes(K);
    const long long* rg = reinterpret_cast<const long long*>(range);

    {
        // x rows (with a token image: only their shared-bound slots are reset), then -- unless the codebook's image is
        // valid -- w rows and the |e|^2 padding
        const int64_t rows = N + (have_w_image ? 0 : kpad);
        const unsigned row_blocks = static_cast<unsigned>((rows + 3) / 4);
        long long zero_blocks = 0;
        for (int r = 0; r < 3; ++r) {
            T5VQ_I_REQUIRE(zero.n[r] >= 0 && (zero.n[r] == 0 || zero.p[r]) &&
                               (reinterpret_cast<uintptr_t>(zero.p[r]) & 15u) == 0,
                           "zero regions must be 16-byte aligned");
            zero_blocks += (zero.n[r] + 4095) / 4096;
        }
        const dim3 pgrid(row_blocks + static_cast<unsigned>(zero_blocks));
        const int nch = static_cast<int>((D + 255) / 256);
        // compute_norms: 0 = xx and ee are inputs, 1 = both outputs, 2 = xx input / ee output, 3 = xx output / ee input
        const int norm_x = (compute_norms == 1 || compute_norms == 3) ? 1 : 0;
        const int norm_w = (compute_norms == 1 || compute_norms == 2) ? 1 : 0;
#define T5VQ_PREP(NCH_)                                                                                          \
    hipLaunchKernelGGL((filter_prep_kernel<NCH_>), pgrid, dim3(256), 0, stream, x, static_cast<long long>(N), xb, \
                       x_inv, w, static_cast<long long>(K), wb, w_inv, xx, ee,                                   \
                       static_cast<long long>(have_w_image ? 0 : kpad), se, static_cast<int>(D), n_list, tshare,   \
                       row_blocks, zero, norm_x, norm_w, order, have_x_image ? 1 : 0, rg,                           \
                       static_cast<long long>(code_offset), static_cast<long long>(code_offset + K))
        if (nch <= 1) { T5VQ_PREP(1); }
        else if (nch <= 2) { T5VQ_PREP(2); }
        else if (nch <= 3) { T5VQ_PREP(3); }
        else if (nch <= 4) { T5VQ_PREP(4); }
        else if (nch <= 8) { T5VQ_PREP(8); }
        else { T5VQ_PREP(16); }
#undef T5VQ_PREP
    }
    T5VQ_I_CHECK_HIP(hipGetLastError());

    // error coefficient of the fp16 dot product against the fp32 chain, relative to |x| |e|:
    // two operand roundings (u = 2^-11 each; elements below 2^-28 of their row maximum fall into fp16's denormals:
    // absolute error 2^-39 of the maximum when rounded, 2^-28 if the pipe flushed them to zero -- the sqrt(D) 2^-26
    // term covers both operands either way), the matrix pipe's fp32 accumulation of the exact
    // products (<= D 2^-22), the chain's own rounding (<= 1.01 D 2^-24); doubled because the distance carries 2 dot;
    // 2 % margin for the fp32 evaluation of the bound itself.  r covers the roundings of the distance expression and
    // of the bound's sums.
    const double u = 1.0 / 2048.0;
    const double c_dot = (2 * u + u * u) + sqrt(static_cast<double>(D)) * (1.0 / 67108864.0) +
                         D * (1.0 / 4194304.0) + 1.01 * D * (1.0 / 16777216.0);
    const float c2 = static_cast<float>(2.0 * c_dot * 1.02);
    const float r = 1.0f / 1048576.0f;  // 2^-20

    const int64_t tok_tiles = (N + TOK_TILE - 1) / TOK_TILE;
    const int64_t code_tiles = (K + CODE_TILE - 1) / CODE_TILE;
    const int64_t tiles_per_split = (code_tiles + splits - 1) / splits;
    const int plain_map = tok_tiles * splits <= 1024 ? 1 : 0;
    const dim3 grid(static_cast<unsigned>(plain_map ? tok_tiles * splits : (tok_tiles + 7) / 8 * 8 * splits));
#define T5VQ_FILTER_ARGS                                                                                              \
    reinterpret_cast<const float*>(xb), reinterpret_cast<const float*>(wb), xx, ee, se, x_inv, w_inv,                 \
        static_cast<int>(N), static_cast<int>(K), static_cast<int>(kpad), static_cast<int>(D / 2),                    \
        static_cast<int>(splits), static_cast<int>(tiles_per_split), plain_map, c2, r, rg,                           \
        static_cast<long long>(code_offset), ubuf, lbuf, cbuf, tshare, order
    if (range)
        hipLaunchKernelGGL((filter_f16_kernel<true>), grid, dim3(THREADS), LDS_BYTES, stream, T5VQ_FILTER_ARGS);
    else
        hipLaunchKernelGGL((filter_f16_kernel<false>), grid, dim3(THREADS), LDS_BYTES, stream, T5VQ_FILTER_ARGS);
#undef T5VQ_FILTER_ARGS
    T5VQ_I_CHECK_HIP(hipGetLastError());
#define T5VQ_COLLECT(L_, R_)                                                                                      \
    hipLaunchKernelGGL((collect_kernel<L_, R_>),                                                                   \
                       dim3(static_cast<unsigned>((N * L_ + COLLECT_THREADS - 1) / COLLECT_THREADS)),                   \
                       dim3(COLLECT_THREADS), 0, stream,                                                             \
                       ubuf, lbuf, cbuf, static_cast<int>(N), static_cast<int>(K), static_cast<int>(scanners),      \
                       exact_keys, rg, static_cast<long long>(code_offset),                                          \
                       static_cast<int>(tiles_per_split * CODE_TILE), cnt, cand,                                     \
                       reinterpret_cast<long long*>(keys), tok_list, n_list, order)
    static_assert(4 * MAX_SPLITS <= 128, "collect_kernel: at most 64 lanes per token, two scanners each");
    if (scanners <= 4) { T5VQ_COLLECT(4, 1); }
    else if (scanners <= 8) { T5VQ_COLLECT(8, 1); }
    else if (scanners <= 16) { T5VQ_COLLECT(16, 1); }
    else if (scanners <= 32) { T5VQ_COLLECT(32, 1); }
    else if (scanners <= 64) { T5VQ_COLLECT(64, 1); }
    else { T5VQ_COLLECT(64, 2); }
#undef T5VQ_COLLECT
    T5VQ_I_CHECK_HIP(hipGetLastError());
    const int64_t refine_blocks = (N * CAND_MAX + 255) / 256 < 2048 ? (N * CAND_MAX + 255) / 256 : 2048;
    hipLaunchKernelGGL(refine_kernel, dim3(static_cast<unsigned>(refine_blocks)), dim3(256), 0, stream, x, w, xx, ee,
                       static_cast<int>(D), rg, static_cast<long long>(code_offset), n_list, cnt, cand,
                       reinterpret_cast<long long*>(keys), n_overflow);
    T5VQ_I_CHECK_HIP(hipGetLastError());
    // tokens the filter could not settle: the fp32 scan kernel itself, on those tokens only
    const int rc = t5vq_internal_scan_listed(x, w, xx, ee, range, N, K, D, code_offset, tok_list, n_list, keys, stream);
    if (rc != T5VQ_OK) return rc;
    return T5VQ_OK;
}

}  // extern "C"
