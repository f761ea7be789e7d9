// conv.hip -- implicit-GEMM convolution on gfx950 MFMA tiles.
//
// Replaces: tf.keras.layers.Conv2D / Conv2DTranspose and the
// tf.pad(...,"REFLECT") in front of them (module.py:210-216,230-264,284-311) and their
// gradients (gen_tape/disc_tape.gradient, model.py:196-197).
//
// Three GEMM kernels, all NHWC with channels padded to 8 so every gather is a 16-byte chunk:
//   conv_gemm<FWD>   y[pixel][k]   = sum_{tap,c} x~[pixel,tap][c] * Wf[k][tap,c]       (gather over x)
//   conv_gemm<DGRAD> dx[pixel][c]  = sum_{tap,k} dy[pixel,tap][k] * Wd[c][tap,k]       (transposed gather; the
//                    stride-2 case is split into stride^2 output-parity classes so each block only visits the
//                    taps that hit its pixels; REFLECT folds the mirrored border gradients in the gather)
//   conv_wgrad       dW[tap,c][k]  = sum_{pixel} x~[pixel,tap][c] * dy[pixel][k]        (reduction over pixels,
//                    operands transposed on the fly with ds_read_b64_tr_b16; split over pixel ranges into f32
//                    slabs that a second kernel sums in fixed order -> deterministic)
// Conv2DTranspose forward IS conv_gemm<DGRAD> of the equivalent conv (plus bias), its data gradient IS
// conv_gemm<FWD>, its weight gradient IS conv_wgrad with the operand roles swapped.
//
// Tile: 256 threads = 4 waves; LDS rows are 128-byte K-slices (64 bf16 / 32 f32) XOR-swizzled on the 16-byte
// chunk index so the ds_read_b128 fragment reads are bank-conflict free; MFMA 16x16x32 bf16 (or 16x16x4 f32 on
// the parity path, consuming the same 16-byte fragments with a k-permutation that A and B share).
// The weight fragment is the MFMA A operand and the pixel fragment the B operand, so the accumulator holds
// D[cout][pixel]: each lane owns 4 consecutive output channels of one pixel = one 8/16-byte NHWC store.
#include <type_traits>
#include "common.h"
#include <stdlib.h>

#ifdef SGG_LAB
#define SGG_ABLATE_OF(a) ((a).ablate)
#else
#define SGG_ABLATE_OF(a) 0      // the shipped kernels carry no ablation branches
#endif

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_BORDER = 2 };   // BORDER: dgrad of the REFLECT border pixels only

struct ConvArgs {
    const char* src;     // FWD: x (N,H,W,C)    DGRAD: dy (N,Ho,Wo,K)
    const char* wmat;    // FWD: [K][R*S*C]     DGRAD: [C][R*S*K]
    const float* bias;   // per destination channel or nullptr
    char* dst;           // FWD: y (N,Ho,Wo,K)  DGRAD: dx (N,H,W,C)
    const char* addend;  // optional tensor added to the result (same layout as dst): the skip-connection gradient
    const char* fold;    // REFLECT DGRAD (v2): pre-folded gather rows of the border pixels [pixel][tap][K], else nullptr
    float* stats;        // halo 3x3 kernels only: per-(image, pixel chunk, channel) partial sums for the instance norm next to
                         // the conv, [N][chunks][DC][2] f32; nullptr = not wanted.  Forward: (sum, sumsq) of the output.
                         // Data gradient: (sum g, sum g*xhat) of the norm backward that consumes dx, which needs:
    const char* nx;      //   that norm's input (same shape as dst)
    const float* nstats; //   its (mean, rstd)[N][DC]
    const float* ngamma; //   gamma, beta [DC]
    const float* nbeta;
    int nact; float nleak;   // and the activation fused behind it
    // Forward, "normalise on load" (NORM variant of the halo 3x3 kernel): src is the RAW output of the conv before the instance
    // norm in front of this conv; the kernel applies relu(x * A + B) (A = gamma * rstd, B = beta - mean * A, from nstats /
    // ngamma / nbeta; second network of a pair: ngamma2 / nbeta2) to the landed halo rows in LDS and writes the normalised
    // tensor -- which the backward pass needs as this conv's weight-gradient operand -- to nout on the way.
    char* nout;
    const float* ngamma2;
    const float* nbeta2;
    float* partial;      // split-K (v2): f32 slabs [ksplit][pdst][DC]
    int ksplit;          // 1 = no split
    const char* wmat2;   // halo 3x3 kernels, two networks on one stacked batch: images >= nsplit use wmat2 / bias2
    const float* bias2;
    int nsplit;          //   (INT_MAX: one network)
    // Generic GEMM kernel, GROUPED launch of two networks' call sites of one shape (sgg_*_group2): grp = 2 doubles the grid's z
    // dimension; the second group reads / writes at these byte offsets from the pointers above and uses wmat2 / bias2.  Each
    // group is exactly the single-network launch (same tiles, same split-K), so results are bit-identical to two launches.
    int grp;
    size_t net_src, net_dst, net_add, net_part;
    int dst_f32;         // halo 3x3 data gradient, mixed mode: dst is f32 (the gradient chain between instance norms keeps f32)
    int addend_f32;      //   ... and so is the addend
    int ablate;          // lab build only (SGG_ABLATE; always 0 and compiled out otherwise): 1 no in-loop DMA, 2 no LDS reads/MFMAs, 3 = 1 + no barrier,
                         // 5 prologue + epilogue only, 6 prologue only (results in DESIGN.md section 7)
    size_t pdst;         // destination pixels (slab stride)
    int N, H, W, C, K, R, S, stride, pad_t, pad_l, Ho, Wo, reflect;
    int act;
    float leak;
};

struct WgradArgs {
    const char* x;       // (N,H,W,C)  gathered operand
    const char* dy;      // (N,Ho,Wo,K)
    float* ws;           // [splits][R*S*C][K] f32 slabs
    int N, H, W, C, K, R, S, stride, pad_t, pad_l, Ho, Wo, reflect;
    int P, pix_per_split;
    FastDiv dHW, dW;     // divide by Ho*Wo, Wo
    // conv_wgrad_glds_kernel only: two NETWORKS of one shape in one launch -- splits [splits_per_net, 2 * splits_per_net) read
    // xb / dyb over the same pixel ranges (each network gets half the blocks and half the slabs of a single call's grid)
    const char* xb; const char* dyb;
    int splits_per_net;
};

// Kernel-selection switches.  The shipped library always returns the defaults below and never reads the environment;
// only the lab build (-DSGG_LAB, `python sg-gan-tf2_amd/build.py --lab` -> libsggan_lab.so, used by tools/) lets
// SGG_* environment variables override them for A/B timing and ablation runs.
const SggConfig& sgg_config() {
#ifdef SGG_LAB
    static const SggConfig cfg = [] {
        SggConfig c;
        auto rd = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
        c.halo3 = rd("SGG_HALO3", c.halo3);                       // 3x3 stride-1 convs through the LDS-resident halo GEMM
        c.s2halo = rd("SGG_S2HALO", c.s2halo);                    // stride-2 data gradient / transposed conv, all parity classes per block
        c.w9 = rd("SGG_W9", c.w9);                                // all-taps 3x3 weight gradient
        c.w9s2 = rd("SGG_W9S2", c.w9s2);                          // ... its stride-2 variant
        c.stem_dgrad_halo = rd("SGG_STEM_DGRAD_HALO", c.stem_dgrad_halo);
        c.n7 = rd("SGG_N7", c.n7);                                // 7x7 64<->3 layers: (channel, tap column) in the GEMM N dimension
        c.wgrad_rowfast = rd("SGG_WGRAD_ROWFAST", c.wgrad_rowfast);
        c.in_fused_maxhw = rd("SGG_IN_FUSED_MAXHW", c.in_fused_maxhw);
        { const char* e = getenv("SGG_CONV_IMPL"); if (e && e[0] == 'r') c.glds = 0; }   // v1 register-staged GEMMs
        c.ablate = rd("SGG_ABLATE", 0);                           // timing experiments: skips work, results are WRONG
        return c;
    }();
    return cfg;
#else
    static const SggConfig cfg;
    return cfg;
#endif
}

static bool use_glds();

template <typename T>
__device__ inline u32x4 chunk_add(const u32x4& a, const u32x4& b) {
    float fa[ET<T>::VEC], fb[ET<T>::VEC];
    ET<T>::unpack(a, fa);
    ET<T>::unpack(b, fb);
#pragma unroll
    for (int i = 0; i < ET<T>::VEC; ++i) fa[i] += fb[i];
    return ET<T>::pack(fa);
}

// candidate padded-grid coordinates that reflect onto h (MirrorPadGrad preimages); returns count, fills j[3]
__device__ inline int reflect_preimages(int h, int H, int p, int* j) {
    int n = 0;
    j[n++] = h + p;
    if (h >= 1 && h <= p) j[n++] = p - h;
    if (h >= H - 1 - p && h <= H - 2) j[n++] = p + 2 * (H - 1) - h;
    return n;
}

template <typename T, int MODE, int BM, int BN, int WGM>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvArgs a) {
    constexpr int VEC = ET<T>::VEC;
    constexpr int ES = (int)sizeof(T);
    constexpr int WGN = 4 / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int PA = BM / 32;                 // pixel-tile chunks per thread
    constexpr int QA = (BN + 31) / 32;          // weight-tile chunks per thread
    static_assert(BM % 32 == 0 && WM % 16 == 0 && WN % 16 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sP = smem;                    // [2][BM][128]
    char* sQ = smem + 2 * BM * 128;     // [2][BN][128]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index as a scalar
    const int wm = wave / WGN, wn = wave % WGN;
    const int st = (MODE == MODE_DGRAD) ? a.stride : 1;      // tap-enumeration step
    const int ph = (MODE == MODE_DGRAD) ? (int)blockIdx.z / a.stride : 0;
    const int pw = (MODE == MODE_DGRAD) ? (int)blockIdx.z % a.stride : 0;
    const int nr = ph < a.R ? (a.R - ph + st - 1) / st : 0;  // taps of this class along r / s
    const int ns = pw < a.S ? (a.S - pw + st - 1) / st : 0;
    const int SC = (MODE == MODE_FWD) ? a.C : a.K;           // reduction channels (source tensor)
    static_assert(MODE == MODE_FWD || MODE == MODE_DGRAD || MODE == MODE_BORDER, "mode");
    const int DC = (MODE == MODE_FWD) ? a.K : a.C;           // destination channels
    const int cpv = SC / VEC;                                // 16-byte chunks per tap
    const int wrow = a.R * a.S * SC;                         // weight row length (elements)

    // ---- destination pixel space of this block ----
    // FWD: all output pixels.  DGRAD: the pixels of parity class (ph,pw).  BORDER: only the pixels whose gradient
    // receives mirrored contributions (rows 1..p, H-1-p..H-2 and the same columns), or every pixel when the
    // image is too small for those bands to be disjoint.
    int hbase = 0, wbase = 0, Hc, Wc;
    const int p2 = 2 * a.pad_t;
    const bool ball = (MODE == MODE_BORDER) && (a.H <= p2 + 1 || a.W <= p2 + 1);
    int Bimg = 0;
    if (MODE == MODE_FWD) { Hc = a.Ho; Wc = a.Wo; }
    else if (MODE == MODE_DGRAD) {
        hbase = ((ph - a.pad_t) % st + st) % st;
        wbase = ((pw - a.pad_l) % st + st) % st;
        Hc = hbase < a.H ? (a.H - hbase + st - 1) / st : 0;
        Wc = wbase < a.W ? (a.W - wbase + st - 1) / st : 0;
    } else { Hc = a.H; Wc = a.W; Bimg = ball ? a.H * a.W : p2 * a.W + (a.H - p2) * p2; }
    const int M = (MODE == MODE_BORDER) ? a.N * Bimg : a.N * Hc * Wc;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= M) return;                                     // whole block outside this class (uniform)

    // m -> (image, destination h, destination w); FWD returns the output pixel (ho, wo)
    auto decode = [&](int m, int& n, int& h, int& w) {
        if (MODE == MODE_BORDER && !ball) {
            n = m / Bimg;
            int q = m - n * Bimg;
            const int p = a.pad_t;
            if (q < p2 * a.W) {
                int hi = q / a.W;
                w = q - hi * a.W;
                h = hi < p ? 1 + hi : a.H - 1 - p + (hi - p);
            } else {
                q -= p2 * a.W;
                int hh = q / p2, wi = q - hh * p2;
                h = hh == 0 ? 0 : (hh == a.H - p2 - 1 ? a.H - 1 : p + hh);
                w = wi < p ? 1 + wi : a.W - 1 - p + (wi - p);
            }
        } else {
            n = m / (Hc * Wc);
            int rem = m - n * (Hc * Wc);
            int hp = rem / Wc;
            h = hbase + st * hp; w = wbase + st * (rem - hp * Wc);
        }
    };

    // per-thread staging rows: row = (tid>>3) + 32*i, chunk column c = tid&7
    const int cc0 = tid & 7;
    int rn[PA], rh[PA], rw[PA];                              // image, and the row's h/w anchor
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        int m = m0 + (tid >> 3) + 32 * i;
        if (m < M) {
            int n, h, w;
            decode(m, n, h, w);
            rn[i] = n;
            if (MODE == MODE_FWD) { rh[i] = h * a.stride - a.pad_t; rw[i] = w * a.stride - a.pad_l; }
            else if (MODE == MODE_DGRAD) { rh[i] = (h + a.pad_t - ph) / st; rw[i] = (w + a.pad_l - pw) / st; }
            else { rh[i] = h; rw[i] = w; }                   // BORDER keeps destination coords; folds in the gather
        } else rn[i] = -1;
    }

    // tap iterator of this thread's chunk column (shared by all its rows)
    int t_cc, t_ri, t_si;
    { int ti = cc0 / cpv; t_cc = cc0 - ti * cpv; t_ri = ns ? ti / ns : nr; t_si = ns ? ti - t_ri * ns : 0; }
    const int ktiles = (nr * ns * cpv + 7) / 8;

    u32x4 regP[PA], regQ[QA];
    bool anyP = false;                                       // BORDER: did this thread load anything for the tile

    auto load_tile = [&]() {
        anyP = false;
        const bool tapok = t_ri < nr;
        const int r = ph + st * t_ri, s = pw + st * t_si;
        // weights
#pragma unroll
        for (int i = 0; i < QA; ++i) {
            int row = (tid >> 3) + 32 * i, dc = n0 + row;
            regQ[i] = zero16();
            if (tapok && row < BN && dc < DC)
                regQ[i] = ld16(a.wmat + ((size_t)dc * wrow + (size_t)(r * a.S + s) * SC + t_cc * VEC) * ES);
        }
        // pixels
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            regP[i] = zero16();
            if (!tapok || rn[i] < 0) continue;
            if (MODE == MODE_FWD) {
                int hi = rh[i] + r, wi = rw[i] + s;
                bool ok = true;
                if (a.reflect) {
                    hi = hi < 0 ? -hi : (hi >= a.H ? 2 * (a.H - 1) - hi : hi);
                    wi = wi < 0 ? -wi : (wi >= a.W ? 2 * (a.W - 1) - wi : wi);
                } else ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                if (ok) regP[i] = ld16(a.src + (((size_t)rn[i] * a.H + hi) * a.W + wi) * SC * ES + t_cc * 16);
            } else if (MODE == MODE_DGRAD) {
                // REFLECT convs take this path too (ho = h + p - r): the mirrored contributions of the border
                // pixels are added by the BORDER launch that follows and overwrites those pixels.
                int ho = rh[i] - t_ri, wo = rw[i] - t_si;
                if ((unsigned)ho < (unsigned)a.Ho && (unsigned)wo < (unsigned)a.Wo)
                    regP[i] = ld16(a.src + (((size_t)rn[i] * a.Ho + ho) * a.Wo + wo) * SC * ES + t_cc * 16);
            } else {                                         // MirrorPadGrad fold: the MIRRORED preimages only
                int jh[3], jw[3];
                int nh = reflect_preimages(rh[i], a.H, a.pad_t, jh);
                int nw = reflect_preimages(rw[i], a.W, a.pad_l, jw);
                bool first = true;
                for (int ia = 0; ia < nh; ++ia)
                    for (int ib = 0; ib < nw; ++ib) {
                        if (ia == 0 && ib == 0) continue;    // the direct term was written by the main launch
                        int ho = jh[ia] - r, wo = jw[ib] - s;
                        if ((unsigned)ho < (unsigned)a.Ho && (unsigned)wo < (unsigned)a.Wo) {
                            u32x4 v = ld16(a.src + (((size_t)rn[i] * a.Ho + ho) * a.Wo + wo) * SC * ES + t_cc * 16);
                            regP[i] = first ? v : chunk_add<T>(regP[i], v);
                            first = false;
                        }
                    }
                anyP |= !first;
            }
        }
        // advance the tap iterator by one K-tile (8 chunks)
        t_cc += 8;
        while (t_cc >= cpv) { t_cc -= cpv; if (++t_si == ns) { t_si = 0; ++t_ri; } }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            int row = (tid >> 3) + 32 * i;
            st16(sP + (buf * BM + row) * 128 + ((cc0 ^ ((row >> 1) & 7)) << 4), regP[i]);
        }
#pragma unroll
        for (int i = 0; i < QA; ++i) {
            int row = (tid >> 3) + 32 * i;
            if (row < BN) st16(sQ + (buf * BN + row) * 128 + ((cc0 ^ ((row >> 1) & 7)) << 4), regQ[i]);
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4, fsw = frow >> 1;   // fragment row / k-chunk / swizzle key

    auto compute_tile = [&](int buf) {
        const char* bP = sP + (buf * BM + wm * WM + frow) * 128;
        const char* bQ = sQ + (buf * BN + wn * WN + frow) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int off = (((fq + 4 * kk) ^ fsw) << 4);
            u32x4 fp[MI], fw[NI];
#pragma unroll
            for (int j = 0; j < MI; ++j) fp[j] = ld16(bP + j * 16 * 128 + off);
#pragma unroll
            for (int i = 0; i < NI; ++i) fw[i] = ld16(bQ + i * 16 * 128 + off);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    if constexpr (sizeof(T) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, fw[i]), __builtin_bit_cast(bf16x8, fp[j]), acc[i][j], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                __uint_as_float(fw[i][e]), __uint_as_float(fp[j][e]), acc[i][j], 0, 0, 0);
                    }
                }
        }
    };

    if (MODE == MODE_BORDER) {
        // mirrored terms touch few (tap, pixel) pairs: skip K-tiles in which no row of the block has a source
        for (int kt = 0; kt < ktiles; ++kt) {
            load_tile();
            if (!__syncthreads_or(anyP ? 1 : 0)) continue;   // also fences the previous tile's LDS reads
            store_tile(0);
            __syncthreads();
            compute_tile(0);
        }
    } else {
        if (ktiles > 0) { load_tile(); store_tile(0); }
        __syncthreads();
        for (int kt = 0; kt < ktiles; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < ktiles) load_tile();
            compute_tile(buf);
            if (kt + 1 < ktiles) store_tile(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds D[cout = 4*fq + e][pixel = frow] of each 16x16 tile ----
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        int m = m0 + wm * WM + j * 16 + frow;
        if (m >= M) continue;
        size_t dpix;
        if (MODE == MODE_FWD) dpix = (size_t)m;
        else {
            int n, h, w;
            decode(m, n, h, w);
            dpix = ((size_t)n * a.H + h) * a.W + w;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int dc = n0 + wn * WN + i * 16 + fq * 4;
            if (dc >= DC) continue;
            float v[4];
            T* o = reinterpret_cast<T*>(a.dst) + dpix * DC + dc;
            if (MODE == MODE_BORDER) {                       // add the mirrored terms to what the main launch wrote
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + (float)o[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[i][j][e] + (a.bias ? a.bias[dc + e] : 0.f);
                    v[e] = act_apply(t, a.act, a.leak);
                    if (a.addend) v[e] += (float)(reinterpret_cast<const T*>(a.addend) + dpix * DC + dc)[e];
                }
            }
            if constexpr (sizeof(T) == 2) {
                bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                *reinterpret_cast<bf16x4*>(o) = pk;
            } else {
                *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// v2 GEMM kernel: direct-to-LDS staging (global_load_lds_dwordx4), single 32 KB LDS stage, ~4 blocks per CU.
//
// Each wave-instruction writes 1 KiB = 8 tile rows x 128 B linearly (LDS dest = wave-uniform base + lane*16), so the
// XOR swizzle is applied on the SOURCE side: lane (row, pos) fetches logical chunk pos ^ swz(row) and the fragment
// reads look chunk j up at position j ^ swz(row) (same involution).  Out-of-range taps / rows read a 16-byte zero
// page instead of branching.  Two LDS stages: the DMA of tile t+1 is in flight while tile t is multiplied.
// Tiles up to 256x256 with 8 waves (one block per CU): 128 FLOP per staged byte keeps the L2->LDS stream under
// what the memory side delivers (PMC: the 128x128 variant was bound there at ~9 TB/s of L2 reads, 93 % hits).
// -------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) uint32_t g_zero_page[4] = {0u, 0u, 0u, 0u};
// debug aid (SGG_ABLATE=9): shader-clock and wall-clock timestamps of one block of the halo GEMM, read by sgg_debug_clocks()
__device__ unsigned long long g_dbg_clk[4];
// lab build, SGG_ABLATE=8: shader-clock stamps of block 0, per wave, at six points of each of 16 mid-loop tiles
// [wave][tile][point]: 0 loop top, 1 refill DMAs issued, 2 all MFMAs issued, 3 vmcnt(0) passed, 4 lgkmcnt(0) passed, 5 barrier passed
__device__ unsigned long long g_dbg_phase[8][16][6];

#ifndef GL_WIDE
#define GL_WIDE 1                                      // epilogue with 16-byte stores (two fragments exchanged between lane rows), bf16 results
#endif
template <typename T, int MODE, int BM, int BN, int WGM, int NW, int BKB, int NS>
__global__ __launch_bounds__(NW * 64) void conv_gemm_glds_kernel(ConvArgs a_in) {
    // grouped launch (a.grp == 2): blockIdx.z = group * classes + parity class; group 1 is the second network's call
    ConvArgs a = a_in;
    const int zclasses = (MODE == MODE_DGRAD) ? a.stride * a.stride : 1;
    const int zgrp = (int)blockIdx.z / zclasses, zcls = (int)blockIdx.z - zgrp * zclasses;
    if (zgrp) {
        a.src += a.net_src; a.dst += a.net_dst; a.wmat = a.wmat2; a.bias = a.bias2;
        if (a.addend) a.addend += a.net_add;
        if (a.partial) a.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(a.partial) + a.net_part);
    }
    constexpr int VEC = ET<T>::VEC;
    constexpr int ES = (int)sizeof(T);
    constexpr int WGN = NW / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int CPR = BKB / 16;               // 16-byte chunks per LDS row (8: 128-byte K-slices, 4: 64-byte)
    constexpr int RPW = 64 / CPR;               // tile rows per 1-KiB wave-instruction
    constexpr int RPP = NW * RPW;               // tile rows staged per pass
    constexpr int PA = BM / RPP;
    constexpr int QA = (BN + RPP - 1) / RPP;
    constexpr int BNR = QA * RPP;               // weight-tile rows incl. padding
    constexpr int STAGE = (BM + BNR) * BKB;     // bytes per LDS stage
    constexpr int LPT = PA + QA;                // DMA instructions per thread per K-tile
    constexpr int KK = BKB / 64;                // 32-element MFMA k-steps per K-tile
    static_assert(BKB == 128 || BKB == 64, "row bytes");
    static_assert(NS >= 2 && (NS - 2) * LPT <= 63, "stages / vmcnt range");
    static_assert(MODE == MODE_FWD || MODE == MODE_DGRAD, "mode");
    static_assert(BM % RPP == 0 && WM % 16 == 0 && WN % 16 == 0 && WGM * WGN == NW, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];   // NS stages of { P [BM][BKB], Q [BNR][BKB] }

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index as a scalar
    const int wm = wave / WGN, wn = wave % WGN;
    const int st = (MODE == MODE_DGRAD) ? a.stride : 1;
    const int ph = (MODE == MODE_DGRAD) ? zcls / a.stride : 0;
    const int pw = (MODE == MODE_DGRAD) ? zcls % a.stride : 0;
    const int nr = ph < a.R ? (a.R - ph + st - 1) / st : 0;
    const int ns = pw < a.S ? (a.S - pw + st - 1) / st : 0;
    const int SC = (MODE == MODE_FWD) ? a.C : a.K;
    const int DC = (MODE == MODE_FWD) ? a.K : a.C;
    const int cpv = SC / VEC;
    const int wrow = a.R * a.S * SC;

    // ---- pixel space of this block ----
    int hbase = 0, wbase = 0, Hc, Wc;
    if (MODE == MODE_FWD) { Hc = a.Ho; Wc = a.Wo; }
    else {
        hbase = ((ph - a.pad_t) % st + st) % st;
        wbase = ((pw - a.pad_l) % st + st) % st;
        Hc = hbase < a.H ? (a.H - hbase + st - 1) / st : 0;
        Wc = wbase < a.W ? (a.W - wbase + st - 1) / st : 0;
    }
    const int M = a.N * Hc * Wc;
    const int tilesN = (DC + BN - 1) / BN;
    // 1-D grid over (M-tile, N-tile) pairs, N fastest, so the N-tiles of one pixel tile run back to back; blocks are
    // remapped so that each XCD (blocks b, b+8, ... share one) walks a CONTIGUOUS range of pixel tiles: the halo rows
    // that neighbouring tiles re-read then hit that XCD's private L2 (speed only; any placement is correct).
    int lid;
    {
        const int b = (int)blockIdx.x, nm = (int)gridDim.x;
        const int q = nm >> 3, rr = nm & 7, xcd = b & 7;
        lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
    }
    const int m0 = (lid / tilesN) * BM, n0 = (lid % tilesN) * BN;
    if (m0 >= M) return;

    auto decode = [&](int m, int& n, int& h, int& w) {
        n = m / (Hc * Wc);
        int rem = m - n * (Hc * Wc);
        int hp = rem / Wc;
        h = hbase + st * hp; w = wbase + st * (rem - hp * Wc);
    };

    // swizzle key of a tile row: 128-byte rows: (row>>1)&7; 64-byte rows: (-(row>>2))&3  (both make every 16-lane
    // ds_read_b128 group hit 16 distinct 16-byte slots of the 256-byte bank line)
    auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : ((-(row >> 2)) & 3); };
    // staging rows of this thread: row = tid/CPR + RPP*i at LDS position tid%CPR; logical chunk = pos ^ swz(row),
    // and swz(row) is the same for every pass i (RPP is a multiple of 16)
    const int srow0 = tid / CPR;
    const int lcc = (tid % CPR) ^ swz(srow0);
    int rn[PA], rh[PA], rw[PA];
    const bool fold = (MODE == MODE_DGRAD) && a.reflect;     // REFLECT data-gradient: MirrorPadGrad terms on the border
    unsigned bmask = 0;                                       // rows of this thread that receive mirrored terms
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        int m = m0 + srow0 + RPP * i;
        if (m < M) {
            int n, h, w;
            decode(m, n, h, w);
            rn[i] = n;
            if (fold) {
                // border pixel -> its row in the pre-folded side tensor (same enumeration as fold_gather_kernel)
                const int p = a.pad_t, p2 = 2 * p;
                const bool all = a.H <= p2 + 1 || a.W <= p2 + 1;
                const bool rowband = (h >= 1 && h <= p) || (h >= a.H - 1 - p && h <= a.H - 2);
                const bool colband = (w >= 1 && w <= p) || (w >= a.W - 1 - p && w <= a.W - 2);
                if (all || rowband || colband) {
                    int q, Bimg;
                    if (all) { Bimg = a.H * a.W; q = h * a.W + w; }
                    else {
                        Bimg = p2 * a.W + (a.H - p2) * p2;
                        if (rowband) q = (h <= p ? h - 1 : p + h - (a.H - 1 - p)) * a.W + w;
                        else {
                            int hh = h == 0 ? 0 : (h == a.H - 1 ? a.H - p2 - 1 : h - p);
                            int wi = w <= p ? w - 1 : p + w - (a.W - 1 - p);
                            q = p2 * a.W + hh * p2 + wi;
                        }
                    }
                    bmask |= 1u << i;
                    rh[i] = n * Bimg + q; rw[i] = 0;
                    continue;
                }
            }
            if (MODE == MODE_FWD) { rh[i] = h * a.stride - a.pad_t; rw[i] = w * a.stride - a.pad_l; }
            else { rh[i] = (h + a.pad_t - ph) / st; rw[i] = (w + a.pad_l - pw) / st; }
        } else rn[i] = -1;
    }

    // split-K (a.ksplit > 1): blockIdx.y selects a contiguous range of K-tiles; partial sums go to f32 slabs
    const int ktot = (nr * ns * cpv + CPR - 1) / CPR;
    const int kper = (ktot + a.ksplit - 1) / a.ksplit;
    const int kt0 = (int)blockIdx.y * kper;
    const int ktiles = SGG_ABLATE_OF(a) >= 5 ? 0 : max(0, min(ktot, kt0 + kper) - kt0);   // ablate 5/6: no main loop
    int t_cc, t_ri, t_si;
    { int q0 = lcc + kt0 * CPR; int ti = q0 / cpv; t_cc = q0 - ti * cpv; t_ri = ns ? ti / ns : nr; t_si = ns ? ti - t_ri * ns : 0; }
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    // Source addresses are kept per tap: the row/weight base pointers are rebuilt only when this thread's chunk
    // column moves to another tap (every SC/(VEC*CPR) K-tiles); within a tap a K-tile costs one 64-bit add per DMA.
    const char* pbase[PA];
    const char* qbase[QA];
    unsigned pok = 0, qok = 0;
    bool tapchg = true;
    auto rebuild = [&]() {
        const bool tapok = t_ri < nr;
        const int r = ph + st * t_ri, s = pw + st * t_si;
        pok = 0; qok = 0;
#pragma unroll
        for (int i = 0; i < QA; ++i) {
            int dc = n0 + srow0 + RPP * i;
            qbase[i] = zero;
            if (tapok && dc < DC) { qbase[i] = a.wmat + ((size_t)dc * wrow + (size_t)(r * a.S + s) * SC) * ES; qok |= 1u << i; }
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            pbase[i] = zero;
            if (!tapok || rn[i] < 0) continue;
            if (MODE == MODE_FWD) {
                int hi = rh[i] + r, wi = rw[i] + s;
                bool ok = true;
                if (a.reflect) {
                    hi = hi < 0 ? -hi : (hi >= a.H ? 2 * (a.H - 1) - hi : hi);
                    wi = wi < 0 ? -wi : (wi >= a.W ? 2 * (a.W - 1) - wi : wi);
                } else ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                if (ok) { pbase[i] = a.src + (((size_t)rn[i] * a.H + hi) * a.W + wi) * SC * ES; pok |= 1u << i; }
            } else if (fold && ((bmask >> i) & 1u)) {
                // MirrorPadGrad: border pixels gather from the side tensor, whose row (pixel, tap) already
                // holds the sum over the mirrored preimages (fold_gather_kernel)
                pbase[i] = a.fold + ((size_t)rh[i] * (a.R * a.S) + (r * a.S + s)) * SC * ES; pok |= 1u << i;
            } else {
                int ho = rh[i] - t_ri, wo = rw[i] - t_si;
                if ((unsigned)ho < (unsigned)a.Ho && (unsigned)wo < (unsigned)a.Wo) {
                    pbase[i] = a.src + (((size_t)rn[i] * a.Ho + ho) * a.Wo + wo) * SC * ES; pok |= 1u << i;
                }
            }
        }
    };
    auto stage_tile = [&](int stg) {
        char* sP = smem + stg * STAGE;
        char* sQ = sP + BM * BKB;
        if (tapchg) { rebuild(); tapchg = false; }
        const int coff = t_cc * 16;
#pragma unroll
        for (int i = 0; i < QA; ++i) {
            const char* src = qbase[i] + (((qok >> i) & 1u) ? coff : 0);
            dma16_to_lds(src, (__attribute__((address_space(3))) void*)(sQ + (wave * RPW + RPP * i) * BKB));
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const char* src = pbase[i] + (((pok >> i) & 1u) ? coff : 0);
            dma16_to_lds(src, (__attribute__((address_space(3))) void*)(sP + (wave * RPW + RPP * i) * BKB));
        }
        t_cc += CPR;
        while (t_cc >= cpv) { t_cc -= cpv; tapchg = true; if (++t_si == ns) { t_si = 0; ++t_ri; } }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const int fswP = swz(wm * WM + frow), fswQ = swz(wn * WN + frow);   // tile bases are multiples of 16: key per lane

    // NS-stage ring: NS-1 tiles are in flight by DMA while one is multiplied; counted vmcnt leaves NS-2 of them
    // outstanding across the (raw) barrier -- a __syncthreads() there would drain the whole queue.
    auto wait_tiles = [&](bool steady) {
        if (steady) sgg_wait_vm<(NS - 2) * LPT>();
        else SGG_WAIT_VM0();
    };
    for (int t = 0; t < NS - 1; ++t)
        if (t < ktiles) stage_tile(t);
    wait_tiles(ktiles >= NS - 1);
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt % NS;
        const bool refill = kt + NS - 1 < ktiles;
        if (refill && (SGG_ABLATE_OF(a) == 0 || SGG_ABLATE_OF(a) == 2)) stage_tile((kt + NS - 1) % NS);
        const char* bP = smem + cur * STAGE + (wm * WM + frow) * BKB;
        const char* bQ = smem + cur * STAGE + BM * BKB + (wn * WN + frow) * BKB;
        if (SGG_ABLATE_OF(a) != 2) {
            // Fragment pipeline: all weight fragments of the tile up front, pixel fragments in groups of GJ that are
            // fetched one group ahead of the MFMAs that consume them, so the ~100-cycle LDS latency hides behind
            // GJ*NI MFMAs instead of stalling every 8 (what hipcc emitted for the plain j-loop).
            constexpr int GJ = MI >= 4 ? 4 : MI;             // pixel fragments per group
            constexpr int GPK = MI / GJ;                      // groups per k-step
            constexpr int NG = KK * GPK;
            u32x4 fw[KK][NI];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int i = 0; i < NI; ++i) fw[kk][i] = ld16(bQ + i * 16 * BKB + (((fq + 4 * kk) ^ fswQ) << 4));
            u32x4 fp[2][GJ];
#pragma unroll
            for (int jj = 0; jj < GJ; ++jj) fp[0][jj] = ld16(bP + jj * 16 * BKB + ((fq ^ fswP) << 4));
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int kk = g / GPK, jb = (g % GPK) * GJ;
                if (g + 1 < NG) {
                    const int kn = (g + 1) / GPK, jn = ((g + 1) % GPK) * GJ;
#pragma unroll
                    for (int jj = 0; jj < GJ; ++jj)
                        fp[(g + 1) & 1][jj] = ld16(bP + (jn + jj) * 16 * BKB + (((fq + 4 * kn) ^ fswP) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);           // keep the prefetch ABOVE this group's MFMAs
#pragma unroll
                for (int jj = 0; jj < GJ; ++jj)
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        if constexpr (sizeof(T) == 2) {
                            acc[i][jb + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, fw[kk][i]), __builtin_bit_cast(bf16x8, fp[g & 1][jj]), acc[i][jb + jj], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[i][jb + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    __uint_as_float(fw[kk][i][e]), __uint_as_float(fp[g & 1][jj][e]), acc[i][jb + jj], 0, 0, 0);
                        }
                    }
            }
        }
        wait_tiles(refill);
        SGG_WAIT_LGKM0();
        if (SGG_ABLATE_OF(a) < 3) __builtin_amdgcn_s_barrier();
    }

    // bias of this lane's 4 output channels per channel tile, fetched once (a per-store load would put a full
    // L2 round trip in front of every one of the NI*MI stores)
    if (SGG_ABLATE_OF(a) == 6) return;
    float bv[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int dc = n0 + wn * WN + i * 16 + fq * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[i][e] = (a.bias && a.ksplit <= 1 && dc < DC) ? a.bias[dc + e] : 0.f;
    }
    // uniform conditions (split-K, activation, addend) are tested once, outside the unrolled store loops
    auto epilogue = [&](auto mode_c) {
        constexpr int EP = decltype(mode_c)::value;               // -1: split-K partial; else 2 * ACT + (addend ? 1 : 0)
        if constexpr (GL_WIDE && EP >= 0 && sizeof(T) == 2 && MI % 2 == 0) {
            // 16-byte stores (see conv3x3_halo_gemm_kernel's epilogue): pixel fragments 2 jp and 2 jp + 1 trade halves between lane
            // rows, a lane then owns 8 consecutive channels of one GEMM row -- half the store instructions, same values
            const int jo = fq & 1, cb = (fq >> 1) * 8;
#pragma unroll
            for (int jp = 0; jp < MI / 2; ++jp) {
                const int m = m0 + wm * WM + (2 * jp + jo) * 16 + frow;
                const bool mok = m < M;
                size_t dpix = (size_t)m;
                if (!(MODE == MODE_FWD || st == 1)) {
                    int n, h, w;
                    decode(mok ? m : 0, n, h, w);
                    dpix = ((size_t)n * a.H + h) * a.W + w;
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int dc = n0 + wn * WN + i * 16 + cb;
                    const bool ok = mok && dc < DC;
                    float v0[4], v1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v0[e] = act_apply_c<EP / 2>(acc[i][2 * jp][e] + bv[i][e], a.leak);
                        v1[e] = act_apply_c<EP / 2>(acc[i][2 * jp + 1][e] + bv[i][e], a.leak);
                    }
                    u32x4 pk;
                    if constexpr (EP & 1) {
                        float ad[8], o[8];
                        ET<bf16>::unpack(ok ? ld16(reinterpret_cast<const bf16*>(a.addend) + dpix * DC + dc) : zero16(), ad);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { row_swap16(v0[e], v1[e]); o[e] = v0[e] + ad[e]; o[4 + e] = v1[e] + ad[4 + e]; }
                        pk = ET<bf16>::pack(o);
                    } else {
                        const bf16x4 p0 = {(bf16)v0[0], (bf16)v0[1], (bf16)v0[2], (bf16)v0[3]}, p1 = {(bf16)v1[0], (bf16)v1[1], (bf16)v1[2], (bf16)v1[3]};
                        const u32x2 q0 = __builtin_bit_cast(u32x2, p0), q1 = __builtin_bit_cast(u32x2, p1);
                        uint32_t a0 = q0[0], a1 = q0[1], b0 = q1[0], b1 = q1[1];
                        row_swap16(a0, b0); row_swap16(a1, b1);
                        pk = (u32x4){a0, a1, b0, b1};
                    }
                    if (ok) st16(reinterpret_cast<bf16*>(a.dst) + dpix * DC + dc, pk);
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            int m = m0 + wm * WM + j * 16 + frow;
            if (m >= M) continue;
            size_t dpix;
            if (MODE == MODE_FWD || st == 1) dpix = (size_t)m;       // stride 1: destination pixel index == GEMM row
            else {
                int n, h, w;
                decode(m, n, h, w);
                dpix = ((size_t)n * a.H + h) * a.W + w;
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int dc = n0 + wn * WN + i * 16 + fq * 4;
                if (dc >= DC) continue;
                if constexpr (EP < 0) {                              // f32 partial; bias/activation are applied by the reducer
                    float* o = a.partial + ((size_t)blockIdx.y * a.pdst + dpix) * DC + dc;
                    *reinterpret_cast<f32x4*>(o) = acc[i][j];
                } else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_apply_c<EP / 2>(acc[i][j][e] + bv[i][e], a.leak);
                    if constexpr (EP & 1) {                          // + skip-connection gradient (module.py:217 backward)
                        const T* ad = reinterpret_cast<const T*>(a.addend) + dpix * DC + dc;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)ad[e];
                    }
                    T* o = reinterpret_cast<T*>(a.dst) + dpix * DC + dc;
                    if constexpr (sizeof(T) == 2) {
                        bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                        *reinterpret_cast<bf16x4*>(o) = pk;
                    } else {
                        *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
                    }
                }
            }
        }
    };
    if (a.ksplit > 1) epilogue(std::integral_constant<int, -1>{});
    else act_dispatch(a.act, [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
        if (a.addend) epilogue(std::integral_constant<int, 2 * ACT + 1>{});
        else epilogue(std::integral_constant<int, 2 * ACT>{});
    });
}

// -------------------------------------------------------------------------------------------------
// 3x3 stride-1 "same" convolution with the INPUT HALO RESIDENT IN LDS (the 18 residual-block convs of the
// generator, module.py:210-216, are this shape at C = K = 256).
//
// conv_gemm_glds_kernel re-fetches a pixel tile once per tap (9x) from L2; at 256x256 tiles that kernel is bound by
// the L2->LDS stream (DMA-only 62 us vs 49 us of MFMA).  Here a block owns 2 output rows x 128
// columns; per 64-channel chunk the 4 x 130 input pixels it needs sit in LDS once (REFLECT / zero padding resolved
// in the DMA's per-lane source address) and the 9 taps read shifted rows of that image, so the pixel operand costs
// 68 KB instead of 288 KB per chunk and only the weight tiles stream per tap (-38 % L2->LDS bytes overall).
// The halo is single-buffered and refilled row by row while other rows are in use: kernel row r touches halo rows
// r and r+1 only, so row 0 of the next chunk is loaded during r=1, row 1 during r=2, rows 2/3 during r=0/1 of the
// next chunk.  K order: chunk-major, tap-minor.  Weight tiles: the same 2-stage DMA ring as above.
// -------------------------------------------------------------------------------------------------
#define H3_TW 128
#ifndef H3_ISSUER_HALF
#define H3_ISSUER_HALF 1                               // which half of the waves (0: waves 0-3, 1: waves 4-7) issues the main loop's weight DMAs
#endif
#ifndef H3_HALO_HALF
#define H3_HALO_HALF 1                                 // ... and which half the halo-row DMAs (the other half measured 3-6 % slower:
#endif                                                 //     the non-issuers' early MFMAs are what covers the issuers' DMA phase)
#ifndef H3_ROT
#define H3_ROT 1                                       // halo rows: 16-byte chunk c of row r sits at position (c + (r & 6)) & 7 (a rotation) instead of c ^ ((r >> 1) & 7)
#endif
#ifndef SGG_NT_ADDEND
#define SGG_NT_ADDEND 0         // 1: the 3x3 halo data gradient reads its skip-gradient addend (its last use) with the streaming cache policy
#endif
#ifndef H3_EARLY_FRAGS
#define H3_EARLY_FRAGS 0 // 1: issuer waves put their first fragment reads of the tile in flight BEFORE the tile's DMA issue (main loop).  OFF: data gradient 3 % slower, forward equal (profiles/r04_persistent_halo_gemm.txt item 8)
#endif
#ifndef H3_LATE_HALO
#define H3_LATE_HALO 0   // 1: halo-row DMAs by the non-issuer waves after their MFMAs, waited for one tile later (main loop).  OFF: 2-3.5 % slower (profiles/r04_persistent_halo_gemm.txt item 7)
#endif
#ifndef H3_NT
#define H3_NT 0          // streaming (nt) LDS-DMA of the halo rows: 1 forward (x), 2 data gradient (dy)
#endif
#ifndef H3_STAGGER
#define H3_STAGGER 0                                   // 1: waves of one half run half a tile behind their SIMD partners (main loop header).  OFF: forward +1 % at best, data gradient spills (profiles/r04_persistent_halo_gemm.txt)
#endif
#ifndef H3_LAG_HALF
#define H3_LAG_HALF 1                                  // which half lags (1: waves 4-7)
#endif
#ifndef H3_PERSIST
#define H3_PERSIST 0                                   // 1: one block per CU walking consecutive tiles, the next tile's prologue issued under the epilogue (kernel header).  OFF: measured flat forward, slower data gradient (profiles/r04_persistent_halo_gemm.txt)
#endif
#ifndef H3_PRIME_FIRST
#define H3_PRIME_FIRST 1                               // persistent form: the next tile's prologue DMAs go out at the HEAD of the epilogue (in front of the addend loads: fewer live registers)
#endif
#ifndef H3_PRIMED_WAIT
#define H3_PRIMED_WAIT 0                               // > 0: at a primed tile's head wait until <= this many vector-memory operations are outstanding (the epilogue's stores) instead of for all
#endif
#ifndef H3_PATCH_ROT
#define H3_PATCH_ROT 0                                 // 1: FOLD column patches laid out so that a patch lane reads the bank slot its halo read would have taken (below): SQ_LDS_BANK_CONFLICT 13 % -> 0, kernel time +0.7-1.5 % -- off
#endif
#ifndef H3_WIDE
#define H3_WIDE 1                                      // epilogue: two pixel fragments' 8-byte (pixel, 4 channels) pieces exchanged between lane rows into 16-byte stores / addend loads
#endif
#ifndef H3_GJ
#define H3_GJ 4                                        // pixel fragments per MFMA group of the main loop (x 4 weight fragments = 16 MFMAs)
#endif
#ifndef H3_NORM_AT
#define H3_NORM_AT 1                                   // NORM: after which MFMA group of the step (0..3) the row transform runs
#endif
// DMA addressing of the halo kernel per source-padding mode (SRC 0 / 1 / 2): 0 = a 64-bit address per lane and piece,
// 1 = uniform base in an SGPR pair + 32-bit lane offset, 2 = buffer descriptor (out-of-range lanes read zeros).  All three give
// the same bits (tests/test_gpu_exact.py passes with each); the defaults are what measured fastest in the step.
#ifndef H3_ADDR0
#define H3_ADDR0 0
#endif
#ifndef H3_ADDR1
#define H3_ADDR1 0
#endif
#ifndef H3_ADDR2
#define H3_ADDR2 1
#endif
#ifndef H3_NORM_ABL
#define H3_NORM_ABL 0                                  // timing experiments only (wrong results): 1 no store of the normalised rows, 2 no in-loop transform, 4 no prologue transform
#endif
#define H3_PITCH 136                                   // halo row pitch in pixels (130 used; multiple of 8 = one DMA)
#define H3_HALO_BYTES (4 * H3_PITCH * 128)
#define H3_PATCH_ROWS 40                               // 2 tile rows x 2 sides x 9 taps = 36, rounded to whole DMAs
#define H3_LDS (H3_HALO_BYTES + 2 * 256 * 128)
#define H3_LDS_FOLD (H3_LDS + 2 * H3_PATCH_ROWS * 128)

// FOLD = data gradient of a REFLECT-padded conv (MirrorPadGrad folded into the gather, module.py:209-216 backward).
// With pad 1 the mirrored terms reach image rows 1 and H-2 and columns 1 and W-2 only:
//   rows   : output row 1 needs (dy[2] + dy[0]) where it would read dy[2], and only that row of the top tile reads
//            that halo slot (same for dy[H-3] + dy[H-1] in the bottom tile), so the slot is filled from a
//            pre-summed "virtual row" (a.fold, second part) -- no change to the fragment reads;
//   columns: the two pixels per tile row in columns 1 / W-2 read their gather row per tap from a small LDS patch
//            (2 x 2 x 9 rows per chunk, from a.fold's first part, which holds the complete mirrored sums for those
//            pixels) through a per-lane address select in the first / last pixel fragment.
// PAIR: two networks of one shape on a stacked batch, images >= a.nsplit take the second weight set (a compile-time flag so
// that the launches of the paired cycle step show under their own name in a kernel trace: they cover twice the images)
// NORM (forward only): the source is the RAW output of the previous conv and the instance norm + ReLU between the two convs
// (module.py:212-214) is applied to the halo rows after they land in LDS -- by the four waves that issue no DMAs, after their
// MFMAs of the step after the row's DMA, one step before its first use -- so the separate apply pass over the tensor (a read
// and a write of every activation) disappears; the normalised interior rows are written to a.nout from the same registers (the
// weight gradient of this conv reads them in the backward pass), a store stream that runs under the MFMAs.  Same arithmetic
// as in_apply_kernel (x * A + B, max 0, RNE to bf16): the results are bit-identical to the two separate calls.
// SRC: how the source tensor is padded -- 0 zeros (every data gradient; the zero-padded forward), 1 REFLECT (forward),
// 2 REFLECT + normalise on load.  A template parameter because the two paddings address their halo DMAs differently (below).
template <int MODE, bool FOLD, int STATS = 0, bool PAIR = false, int SRC = 0>   // STATS: 0 none, 1 forward norm sums, 2 backward norm sums
__global__ __launch_bounds__(512) void conv3x3_halo_gemm_kernel(ConvArgs a) {
    constexpr bool NORM = SRC == 2;
    static_assert(SRC == 0 || (MODE == MODE_FWD && !FOLD), "REFLECT sources / normalise-on-load exist for the forward conv only");
    constexpr int BN = 256, WGM = 2, WGN = 4, WM = 128, WN = 64, MI = 8, NI = 4, BKB = 128, KK = 2;
    constexpr int QA = 4;                              // weight rows per thread per tile (8 waves x 8 rows x 4)
    constexpr int RPP = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sH = smem;                                   // halo [4][H3_PITCH][128 B]
    char* sB = smem + H3_HALO_BYTES;                   // 2 x [256][128 B]
    char* sPatch = smem + H3_LDS;                      // FOLD: 2 x [H3_PATCH_ROWS][128 B]
    typedef __attribute__((address_space(3))) char lds_char;   // DMA destinations as LDS-space pointers from the start
    lds_char* const lH = (lds_char*)smem;
    lds_char* const lB = lH + H3_HALO_BYTES;
    lds_char* const lPatch = lH + H3_LDS;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index as a scalar
    const int wm = wave / WGN, wn = wave % WGN;
    const int SC = (MODE == MODE_FWD) ? a.C : a.K;     // source channels (GEMM K per tap)
    const int DC = (MODE == MODE_FWD) ? a.K : a.C;
    const int wrow = 9 * SC;
    const int nchunk = SC >> 6;

    const int tilesN = (DC + BN - 1) / BN;
    const int tilesW = a.W / H3_TW, tilesH = a.H >> 1;
    int lid;
    {
        const int b = (int)blockIdx.x, nm = (int)gridDim.x;
        const int q = nm >> 3, rr = nm & 7, xcd = b & 7;
        lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
    }
    // PERSISTENT form (H3_PERSIST): the grid is one block per CU and a block walks `tpb` CONSECUTIVE tiles (rows of one image:
    // neighbouring halos, one weight set); the next tile's prologue -- four halo rows, patch, weight tile 0: ~100 KB per block --
    // is issued under the current tile's epilogue (between its addend loads and its stores), so only the first tile of a block
    // pays for an exposed prologue and there is no second wave of block launches.  Same tiles, same arithmetic per tile.
    struct Tile { int n0, img, h0, w0, tw, th; const char* wmat_n; bool hasL, hasR; };
    auto tile_of = [&](int id) {
        Tile c;
        c.n0 = (id % tilesN) * BN;
        int mt = id / tilesN;
        c.tw = mt % tilesW; mt /= tilesW;
        c.th = mt % tilesH;
        c.img = mt / tilesH;
        c.h0 = c.th * 2; c.w0 = c.tw * H3_TW;
        c.wmat_n = PAIR && c.img >= a.nsplit ? a.wmat2 : a.wmat;
        c.hasL = FOLD && c.tw == 0; c.hasR = FOLD && c.tw == tilesW - 1;
        return c;
    };
    constexpr int ADDR = SRC == 0 ? H3_ADDR0 : (SRC == 1 ? H3_ADDR1 : H3_ADDR2);
    constexpr bool PERSIST = H3_PERSIST && SRC != 2 && ADDR == 0 && STATS != 2;
    const int total_tiles = tilesN * tilesW * tilesH * a.N;
    const int tpb = PERSIST ? (total_tiles + (int)gridDim.x - 1) / (int)gridDim.x : 1;
    bool primed = false;                               // this tile's prologue DMAs were issued under the previous tile's epilogue
  for (int it = 0; it < tpb; ++it) {
    const int tile_id = PERSIST ? lid * tpb + it : lid;
    if (PERSIST && tile_id >= total_tiles) break;
    const Tile cur = tile_of(tile_id);
    const int n0 = cur.n0, tw = cur.tw, th = cur.th, img = cur.img, h0 = cur.h0, w0 = cur.w0;
    const char* const wmat_n = cur.wmat_n;
    const bool dbg_clk = SGG_ABLATE_OF(a) == 9 && lid == 0 && tid == 0;
    const int abl = SGG_ABLATE_OF(a) >= 8 ? 0 : SGG_ABLATE_OF(a);    // 8, 9 = full kernel + clock stamps
    if (dbg_clk) { g_dbg_clk[0] = clock64(); g_dbg_clk[1] = wall_clock64(); }

    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    // ---- halo DMA: one wave-instruction = 8 halo pixels x 128 B; a halo row is 17 of them (wave, wave+8, wave+16)
    const int hpos = lane & 7, hsub = lane >> 3;
    constexpr bool mirror = SRC != 0;                                      // (the data gradient always zero-pads dy)
    // DMA addressing (ADDR), three forms measured against each other in the step (profiles/r03_ab_dma_addressing.txt):
    // 0 -- a 64-bit address per lane and piece; zero-padded sources select against the zero page.  ~12 instructions per piece on
    //      the issuer waves, and hipcc keeps every piece's lane mask (an SGPR pair) and base (a VGPR pair) live across the main
    //      loop: ~90 SGPRs parked in VGPR lanes in the data gradient, ~50 v_readlane per step to get them back.
    // 1 -- a uniform base in an SGPR pair + a 32-bit lane offset (REFLECT sources / weight tiles): ~16 VGPRs fewer.
    // 2 -- buffer descriptors: base and size in four SGPRs per operand, the scalar part of a piece's address in the
    //      instruction's soffset, the lane part one loop-invariant VGPR per piece position, lanes outside the buffer read ZEROS:
    //      no zero page, no masks, no vector instruction per piece (issue part of a step: 201 -> 29 VALU, no v_readlane).
    // Form 0 is the FASTEST: paired forward 0.495 of peak against 0.484 (form 1) and 0.473 -> 0.454 (form 2, another box);
    // paired data gradient 0.451 against 0.415 (form 1 for the weight tiles) and 0.434 -> 0.422 (form 2).  An LDS-DMA piece
    // costs the issuing wave 60-180 cycles whatever surrounds it (the CU's vector-memory path into LDS), so the address
    // arithmetic of form 0 runs in time the wave would wait anyway, and the denser issue of forms 1 / 2 only bunches the LDS
    // writes against the other waves' fragment reads.  Only the normalise-on-load variant uses form 1: it needs the registers.
    // (What did pay is the padding mode as a template parameter: 0.472 -> 0.495.)
    static_assert(ADDR != 1 || SRC != 0, "the SGPR-base form has no zero padding");
    constexpr bool SADDR = ADDR == 1, BUFA = ADDR == 2;
    const char* vrows = FOLD ? a.fold + (size_t)a.N * a.H * 18 * SC * 2 : nullptr;   // [N][2][W][SC] after the patches
    const uint32_t rowbytes = (uint32_t)(a.W * SC * 2);
    sgg_rsrc_t rsS, rsV, rsW;                              // this image of the source, its two virtual rows (FOLD), this tile's weight rows
    if constexpr (BUFA) {
        rsS = sgg_make_rsrc(a.src + (size_t)img * a.H * rowbytes, (uint32_t)a.H * rowbytes);
        rsV = sgg_make_rsrc(FOLD ? vrows + (size_t)img * 2 * rowbytes : a.src, FOLD ? 2 * rowbytes : 0u);
    }
    // (Walking the channel chunks in a per-block rotated order -- so that the blocks of one XCD do not all want the
    // same weight tile at the same moment -- measured 1-2 % slower: first-touch L2 misses are not what the tiles wait for.)
    // (vw, nw): this wave acts as issuer vw of nw -- all 8 waves in the prologue, 4 issuer waves in the main loop
    // (ln: the lane id the per-lane parts of the addresses are derived from -- `lane` in the main loop, where they are loop
    // invariants hipcc keeps in registers; an OPAQUE copy for the prologues, whose lane masks and offsets would otherwise be
    // hoisted out of the persistent tile loop and stay live -- as spilled SGPRs -- across every main loop)
    auto load_halo_row = [&](const Tile& c, int k, int chunk, int vw, int nw, int ln) {
        const int hpos = ln & 7, hsub = ln >> 3;
        const int h0 = c.h0, w0 = c.w0, img = c.img;          // (the tile being LOADED: the next one under an epilogue)
        int hi = h0 - 1 + k;
        bool rowok = true;
        if (mirror) hi = hi < 0 ? -hi : (hi >= a.H ? 2 * (a.H - 1) - hi : hi);
        else rowok = (unsigned)hi < (unsigned)a.H;
        const char* rowp = a.src + ((size_t)img * a.H + (rowok ? hi : 0)) * a.W * SC * 2 + chunk * 128;
        // BUFA: the row's descriptor and scalar offset -- the image, one of the two virtual rows (FOLD), or an EMPTY buffer for a
        // row above / below the image (every lane out of range: zeros) -- so that each piece is the same single instruction
        sgg_rsrc_t rs = rsS;
        uint32_t soff = (uint32_t)(rowok ? hi : 0) * rowbytes + chunk * 128;
        if (FOLD) {
            if (h0 == 0 && k == 3) { rowp = vrows + ((size_t)img * 2 + 0) * a.W * SC * 2 + chunk * 128; rs = rsV; soff = chunk * 128; }
            if (h0 == a.H - 2 && k == 0) { rowp = vrows + ((size_t)img * 2 + 1) * a.W * SC * 2 + chunk * 128; rs = rsV; soff = rowbytes + chunk * 128; }
        }
        if (!rowok) rs[2] = 0;
        // (the issuer index once more through an opaque asm, for the SCALAR parts of a piece -- LDS destination, soffset: as
        // loop invariants hipcc parks them in VGPR lanes, 90 spilled SGPRs and ~50 v_readlane per step in the data gradient)
        int vws = vw;
        if (BUFA) asm volatile("" : "+s"(vws));
#pragma unroll
        for (int qi = 0; qi < 5; ++qi) {
            const int q = vw + nw * qi, qs = vws + nw * qi;
            if (q >= 17) break;
            const int hp = q * 8 + hsub;                                   // halo column 0..135
            int wi = w0 - 1 + hp;
            bool ok = rowok && hp < H3_TW + 2;
            if (mirror) wi = wi < 0 ? -wi : (wi >= a.W ? 2 * (a.W - 1) - wi : wi);
            else ok = ok && (unsigned)wi < (unsigned)a.W;
            // Swizzle of the halo image.  A ds_read_b128 is serviced in four groups of 16 lanes that pair k-quarter fq = 0 (2) of
            // eight fragment rows with fq = 1 (3) of the other eight; per half bank line (row parity) a group therefore reads
            // chunk c of rows 2(m0 + {0,1,6,7}) and chunk c + 1 of rows 2(m0 + {2,3,4,5}), where m0 moves with the tap column
            // (fragment base = halo column + 0 / 1 / 2).  With the XOR key (r >> 1) & 7 those eight positions are distinct only
            // for even m0 -- the 31-34 % conflict replays of round 2 (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) on two tap columns
            // of three.  Rotating instead, position = (c + 2 * (r >> 1)) & 7, gives 2 m0 + {0,2,4,6} and 2 m0 + 1 + {4,6,0,2}:
            // all evens and all odds, distinct for EVERY m0.  The DMA writes LDS linearly, so the rotation goes on the source
            // address: position hpos of row r holds chunk (hpos - (r & 6)) & 7.
            const int hrow_ = k * H3_PITCH + hp;
            const int schunk = H3_ROT ? ((hpos - (hrow_ & 6)) & 7) : (hpos ^ ((hrow_ >> 1) & 7));
            __attribute__((address_space(3))) void* const ldst = (__attribute__((address_space(3))) void*)(lH + (k * H3_PITCH + qs * 8) * 128);
            if (BUFA) {
                // lane part: a loop invariant per piece position (H3_PITCH is a multiple of 8: the rotation does not depend on k);
                // REFLECT: every halo pixel is an image pixel (the six filler columns read mirrored pixels nobody uses)
                const bool colok = mirror || (hp < H3_TW + 2 && (unsigned)wi < (unsigned)a.W);
                const uint32_t voff = colok ? (uint32_t)(wi * SC * 2 + (schunk << 4)) : SGG_BUF_OOB;
                dma16_buf_to_lds(voff, rs, soff, ldst);
            } else if (SADDR) {
                dma16_to_lds_s(rowp, (uint32_t)(wi * SC * 2 + (schunk << 4)), ldst);
            } else {
                const char* src = ok ? rowp + (size_t)wi * SC * 2 + (schunk << 4) : zero;
                if (H3_NT & (MODE == MODE_FWD ? 1 : 2)) dma16_to_lds_nt(src, ldst); else dma16_to_lds(src, ldst);
            }
        }
    };

    // ---- FOLD: column patch, row pe = (tile row * 2 + side) * 9 + tap (halo tap order), 5 DMAs by waves 0..4
    const bool hasL = cur.hasL, hasR = cur.hasR;
    auto load_patch = [&](const Tile& c, int chunk, int vw, int nw, int ln) {
        if (!FOLD) return;
        const int hpos = ln & 7, hsub = ln >> 3;
        const int h0 = c.h0, img = c.img;
        const bool hasL = c.hasL, hasR = c.hasR;
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const int pw = vw + nw * qi;                                   // patch DMA 0..4
            if (pw >= H3_PATCH_ROWS / 8) break;
            const int pe = pw * 8 + hsub;
            int tap_h, side, tr, schunk;
            if (H3_PATCH_ROT && H3_ROT) {
                // Patch row of (tile row tr, tap row r): 6 LDS rows, even ones for the entries whose halo read sits in an even halo
                // row -- (L, sx 1), (R, sx 0), (R, sx 2) -- odd ones for (L, sx 0), (L, sx 2), (R, sx 1); chunk c at position
                // (c + rot) & 7 with rot = the halo rotation of the row the lane would have read: L (1 + sx) & 6, R (14 + sx) & 6.
                // The patch lane of a fragment read then hits exactly the bank slot (row parity, position) of the halo read it
                // replaces, and the ds_read_b128 stays conflict-free (with the XOR-keyed patch rows 13 % of the data gradient's LDS
                // cycles were conflict replays: at W = 128 every tile is both a left and a right edge).
                const int grp = pe / 6, slot = pe - 6 * grp;
                side = (0x34 >> slot) & 1;
                const int sx = (0x681 >> (2 * slot)) & 3;
                tr = grp / 3;
                tap_h = (grp - 3 * tr) * 3 + sx;
                schunk = (hpos - ((0x6c90 >> (3 * (side * 3 + sx))) & 7)) & 7;
            } else {
                tap_h = pe % 9; side = (pe / 9) & 1; tr = pe / 18;
                schunk = hpos ^ ((pe >> 1) & 7);
            }
            const bool ok = pe < 36 && (side ? hasR : hasL);
            // a.fold first part: [N][H][2 sides][9 weight taps][SC]; halo tap t pairs with weight tap 8 - t
            const char* src = ok ? a.fold + (((((size_t)img * a.H + h0 + tr) * 2 + side) * 9 + (8 - tap_h)) * SC + chunk * 64) * 2 + (schunk << 4)
                                 : zero;
            dma16_to_lds(src, (__attribute__((address_space(3))) void*)(lPatch + ((chunk & 1) * H3_PATCH_ROWS + pw * 8) * 128));
        }
    };

    // ---- weight-tile DMA: 32 wave-instructions of 8 rows x 128 B per tile; instruction d covers rows 8d .. 8d+7, lane l row
    // 8d + l/8 at chunk position l%8 (swizzled by the row: key = (l/16 + 4(d&1)) & 7, i.e. the d-even key with bit 2 flipped)
    // One per-lane offset for the whole kernel (row within the 8-row group and the swizzled chunk; d & 1 == vw & 1 because nw is
    // even); DC is a multiple of 8, so an 8-row group is inside the matrix or outside it as a whole: a scalar test.
    const int wl = lane >> 3;
    const int wsw = ((lane & 7) ^ ((lane >> 4) & 7)) << 4;
    const uint32_t wvoff0 = (uint32_t)(wl * wrow * 2 + wsw), wvoff1 = (uint32_t)(wl * wrow * 2 + (wsw ^ 64));
    if constexpr (BUFA) rsW = sgg_make_rsrc(wmat_n + (size_t)n0 * wrow * 2, (uint32_t)((DC - n0 < BN ? DC - n0 : BN) * wrow * 2));
    auto load_w = [&](const Tile& c, int stg, int chunk, int tap, int vw, int nw, int ln) {
        const int wl = ln >> 3;
        const int wsw = ((ln & 7) ^ ((ln >> 4) & 7)) << 4;
        const uint32_t wvoff0 = (uint32_t)(wl * wrow * 2 + wsw), wvoff1 = (uint32_t)(wl * wrow * 2 + (wsw ^ 64));
        const int n0 = c.n0;
        const char* const wmat_n = c.wmat_n;
        const char* const tbase = wmat_n + ((size_t)n0 * wrow + tap * SC + chunk * 64) * 2;
        const uint32_t wvoff = (vw & 1) ? wvoff1 : wvoff0;
        lds_char* sQ = lB + stg * (256 * 128);
        int vws = vw;
        if (BUFA) asm volatile("" : "+s"(vws));          // (see load_halo_row)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = vw + nw * j, ds = vws + nw * j;
            if (d >= 32) break;
            __attribute__((address_space(3))) void* const ldst = (__attribute__((address_space(3))) void*)(sQ + ds * 8 * 128);
            if (BUFA) {
                // rows past DC: the descriptor ends at the matrix (or at the tile), and so that the outcome does not hang on how
                // the range check treats soffset, such a group is asked for with an out-of-range LANE offset
                const uint32_t so = (uint32_t)(ds * 8 * wrow * 2 + (tap * SC + chunk * 64) * 2);
                if (n0 + ds * 8 < DC) dma16_buf_to_lds(wvoff, rsW, so, ldst);
                else dma16_buf_to_lds(SGG_BUF_OOB, rsW, 0u, ldst);
            } else if (!SADDR) {
                const bool ok = n0 + d * 8 + wl < DC;
                const char* src = ok ? wmat_n + (size_t)(n0 + wl) * wrow * 2 + d * ((size_t)8 * wrow * 2) + (tap * SC + chunk * 64) * 2 + (wsw ^ ((d & 1) << 6)) : zero;
                dma16_to_lds(src, ldst);
            } else if (n0 + d * 8 < DC) dma16_to_lds_s(tbase + (size_t)d * 8 * wrow * 2, wvoff, ldst);
            else dma16_to_lds(zero, ldst);
        }
    };

    // ---- NORM: halo row k (just landed, chunk `chunk` of the source channels) -> relu(x * A + B) in place.  A lane owns ONE
    // 16-byte channel group (8 channels: its A / B stay in registers for the row) and walks the row's pixels, 8 per wave pass;
    // rows 1 and 2 are this tile's own image rows: their 128 interior pixels also go to a.nout (whole 128-byte lines).
    float* const sNA = reinterpret_cast<float*>(smem + (FOLD ? H3_LDS_FOLD : H3_LDS));   // A[SC], then B[SC]
    auto norm_row = [&](int k_, int chunk, int vw, int nw) {
        if constexpr (NORM) {
            // (k through an opaque asm: the row's LDS / global addresses are loop invariants otherwise, and hipcc keeps all of
            // them -- four rows x five pieces -- live across the main loop, next to 128 accumulators: scratch spills)
            int k = k_;
            asm volatile("" : "+s"(k));
            const int c8 = lane & 7;
            float A[8], B[8];
            {
                const float* pa = sNA + chunk * 64 + c8 * 8;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(pa), a1 = *reinterpret_cast<const f32x4*>(pa + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(pa + SC), b1 = *reinterpret_cast<const f32x4*>(pa + SC + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { A[e] = a0[e]; A[4 + e] = a1[e]; B[e] = b0[e]; B[4 + e] = b1[e]; }
            }
            const bool own_row = k == 1 || k == 2;
            char* const orow = a.nout + ((((size_t)img * a.H + h0 + k - 1) * a.W + w0) * SC + chunk * 64 + c8 * 8) * 2;
            auto piece_ptr = [&](int q) -> char* {
                const int hrow_ = k * H3_PITCH + q * 8 + hsub;
                const int pos = H3_ROT ? ((c8 + (hrow_ & 6)) & 7) : (c8 ^ ((hrow_ >> 1) & 7));
                return sH + hrow_ * 128 + (pos << 4);
            };
            // up to five pieces per lane and row (17 pixel groups over 4 waves), in batches of <= 3 with the batch's reads issued
            // together: one LDS latency per batch, and no more than 12 staging registers (the accumulators leave little room)
#pragma unroll
            for (int b0 = 0; b0 < 5; b0 += 3) {
                u32x4 v[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int q = vw + nw * (b0 + u);
                    if (b0 + u < 5 && q < 17) v[u] = ld16(piece_ptr(q));
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int q = vw + nw * (b0 + u);
                    if (b0 + u >= 5 || q >= 17) continue;
                    const int hp = q * 8 + hsub;
                    float xv[8], o[8];
                    ET<bf16>::unpack(v[u], xv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float t = xv[e] * A[e] + B[e]; o[e] = t > 0.f ? t : 0.f; }
                    const u32x4 pk = ET<bf16>::pack(o);
                    st16(piece_ptr(q), pk);
                    if (!(H3_NORM_ABL & 1) && own_row && hp >= 1 && hp <= H3_TW) st16(orow + (size_t)(hp - 1) * SC * 2, pk);
                }
            }
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const int fswQ = ((wn * WN + frow) >> 1) & 7;

    if constexpr (NORM) {                                  // the norm's affine form per source channel of this image
        const float* const gm = PAIR && img >= a.nsplit ? a.ngamma2 : a.ngamma;
        const float* const bt = PAIR && img >= a.nsplit ? a.nbeta2 : a.nbeta;
        for (int c = tid; c < SC; c += 512) {
            const float mu = a.nstats[((size_t)img * SC + c) * 2], rs = a.nstats[((size_t)img * SC + c) * 2 + 1];
            const float A = gm[c] * rs;
            sNA[c] = A;
            sNA[SC + c] = bt[c] - mu * A;
        }
    }
    // prologue: whole halo of chunk 0 + weight tile 0
    if (!primed) {
        int lop = lane;
        if (PERSIST) asm volatile("" : "+v"(lop));
#pragma unroll
        for (int k = 0; k < 4; ++k) load_halo_row(cur, k, 0, wave, 8, lop);
        load_patch(cur, 0, wave, 8, lop);
        load_w(cur, 0, 0, MODE == MODE_FWD ? 0 : 8, wave, 8, lop);
    }
    // (primed: the DMAs are older than the previous tile's stores, and vector-memory operations complete in issue order --
    // waiting until at most the stores issued after them are outstanding leaves those in flight under this tile's first steps)
    if (PERSIST && primed && H3_PRIMED_WAIT > 0) sgg_wait_vm<H3_PRIMED_WAIT>(); else SGG_WAIT_VM0();
    __builtin_amdgcn_s_barrier();
    if constexpr (NORM && !(H3_NORM_ABL & 4)) {
        // the tile's own rows (1, 2: stored to a.nout) by the waves that issue no DMAs in the main loop, the neighbours' rows by
        // the DMA issuers: those wait for vmcnt(0) at the end of every step and would sit out the stores' round trip to memory
        // there (with the rows dealt to all eight waves the prologue cost 4.7 us per tile)
        if ((wave >> 2) != H3_HALO_HALF) { norm_row(1, 0, wave & 3, 4); norm_row(2, 0, wave & 3, 4); }
        else { norm_row(0, 0, wave & 3, 4); norm_row(3, 0, wave & 3, 4); }
        SGG_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
    }

    const int ntiles = abl >= 5 ? 0 : nchunk * 9;
    int chunk = 0, tap = 0;                           // of the tile being multiplied
#ifdef SGG_LAB
    const bool stamp = SGG_ABLATE_OF(a) == 8 && lid == 0 && lane == 0;
#define H3_STAMP(pt) do { if (stamp && t >= 8 && t < 24) g_dbg_phase[wave][t - 8][pt] = clock64(); } while (0)
#else
#define H3_STAMP(pt) do {} while (0)
#endif
    // STAGGER (H3_STAGGER): one half of the waves -- one wave of every SIMD pair -- runs half a tile behind the other.  Both halves
    // run the same tiles between the same barriers and read a tile's fragments from LDS inside the tile (so every LDS hazard is the
    // one-barrier-per-tile scheme's), but the lagging half multiplies k-step 1 of a tile AFTER the tile's barrier, from registers:
    // when the barrier opens it has 32 MFMAs ready while its partner waits for LDS, and near the end of the tile its k-step 0
    // finishes early and the partner has the pipe.  Same MFMAs into the same accumulators in the same order: bit-identical.
    // Two copies of the loop (the carried fragments must not count against the other half's registers).
    constexpr bool STAG = H3_STAGGER && !NORM && STATS != 2;     // (the STATS == 2 build of the staggered form aborted on the GPU: fenced off, not debugged)
    auto main_loop = [&](auto lag_c) {
        constexpr bool LAG = decltype(lag_c)::value;
        u32x4 cw[NI], cp[MI];                            // LAG: k-step 1's weight / pixel fragments of the tile before
        // LATE_HALO: the halo rows (and patches) are issued by the waves that issue no weight DMAs, AFTER their MFMAs -- in the ~1 200
        // cycles they otherwise wait at the barrier for their SIMD partners, whose serial path (weight DMAs, then 64 MFMAs) paces the
        // tile -- and waited for one tile LATER with a counted vmcnt (a row is first read three or more tiles after its issue), so
        // that their landing time is on nobody's path.  (Round 2 tried the placement with a full drain in the same tile: 2-3 % slower.)
        constexpr bool LATE_HALO = H3_LATE_HALO && !NORM && !LAG && !STAG;
        auto issue_halo = [&](int vw) -> int {            // returns the number of halo rows issued this tile (0 .. 2); row 0 brings the patch
            int rows = 0;
            if (tap == 0 && chunk > 0) { load_halo_row(cur, 2, chunk, vw, 4, lane); ++rows; }
            if (tap == 3) {
                if (chunk > 0) { load_halo_row(cur, 3, chunk, vw, 4, lane); ++rows; }
                if (chunk + 1 < nchunk) { load_halo_row(cur, 0, chunk + 1, vw, 4, lane); load_patch(cur, chunk + 1, vw, 4, lane); ++rows; }
            }
            if (tap == 6 && chunk + 1 < nchunk) { load_halo_row(cur, 1, chunk + 1, vw, 4, lane); ++rows; }
            return rows;
        };
        for (int t = 0; t < ntiles; ++t) {
            H3_STAMP(0);
            if constexpr (LAG) {
                // k-step 1 of the tile before, from the fragments carried across its barrier: the matrix pipe has work the moment the
                // barrier opens, while the SIMD partner (not lagging) waits for this tile's first fragments to come back from LDS
                if (t > 0) {
#pragma unroll
                    for (int j = 0; j < MI; ++j)
#pragma unroll
                        for (int i = 0; i < NI; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cw[i]), __builtin_bit_cast(bf16x8, cp[j]), acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);           // (the carried fragments are dead from here on: nothing of the DMA issue above them)
            }
            // refill: next weight tile, and the halo rows whose slots are free (see header).  ISSUER WAVES: the CU's vector-memory
            // path takes ~16 cycles per 1 KB DMA instruction, so the ~40 of a tile keep it busy for ~640 cycles; with every wave
            // issuing its share at the head of the tile, all eight sat in issue stalls for that long with the matrix pipes idle
            // (tools/halo_phases.py).  Now waves 4-7 issue ALL the DMAs while waves 0-3 -- their SIMD partners -- go straight to
            // their MFMAs.  (Not the same as splitting each wave's issue in time: waves 4-7 issuing their own share half way
            // through the tile measured 7 % slower; waves 0-3 issuing the halo rows, or 2-4 of the 8 weight DMAs per SIMD pair,
            // AFTER their MFMAs -- in the ~1 000 cycles they wait at the barrier -- measured 2-3 % slower.)
            // EARLY_FRAGS: an issuer wave first puts its OWN first fragment reads (4 weight + GJ pixel fragments of this tile, landed since
            // the last barrier) in flight and issues its DMAs behind them, so that its first MFMA group does not start with an LDS round
            // trip after ~1 000 cycles of DMA issue -- the issuer's serial path paces the tile.
            constexpr bool EARLY_FRAGS = H3_EARLY_FRAGS && !LAG && !NORM;
            auto issue_dmas = [&]() {
                if (abl == 0 || abl == 2) {
                    const int vw = wave & 3;
                    if ((wave >> 2) == H3_ISSUER_HALF) {           // the weight tile: 8 DMA instructions per issuer wave and tile
                        int ntap = tap + 1, nchk = chunk;
                        if (ntap == 9) { ntap = 0; ++nchk; }
                        if (t + 1 < ntiles) load_w(cur, (t + 1) & 1, nchk, MODE == MODE_FWD ? ntap : 8 - ntap, vw, 4, lane);
                    }
                    if (!LATE_HALO && (wave >> 2) == H3_HALO_HALF) issue_halo(vw);   // the halo rows (17 instructions each, ~1 row per tile on average)
                }
            };
            if (!EARLY_FRAGS || abl == 2) issue_dmas();
            H3_STAMP(1);
            const int r = tap / 3, sx = tap - 3 * r;
            // halo offset (r, sx) pairs with weight tap (r, sx) forward and with the flipped tap (2-r, 2-sx) = 8 - tap in
            // the data gradient, so both modes walk the halo rows in the same order (the refill schedule relies on it)
            const int hr = wm + r;                                             // halo row of this wave's output row
            const int hc = frow + sx;                                          // halo column of fragment 0
            const int fswP = H3_ROT ? ((hr * H3_PITCH + hc) & 6) : (((hr * H3_PITCH + hc) >> 1) & 7);   // same for every fragment (16 | fragment step)
            const char* bP = sH + (hr * H3_PITCH + hc) * 128;
            const char* bQ = sB + (t & 1) * (256 * 128) + (wn * WN + frow) * 128;
            // FOLD: the lanes holding pixel column 1 (fragment 0) / W-2 (fragment MI-1) read their patch row instead
            constexpr bool PROT = H3_PATCH_ROT && H3_ROT;
            // PROT: row (wm * 3 + r) * 6 + slot, slot L = {1, 0, 3}[sx], R = {2, 5, 4}[sx]; rotation L = {0, 2, 2}[sx], R = {6, 6, 0}[sx] (load_patch)
            const int peL = PROT ? (wm * 3 + r) * 6 + ((0x31 >> (2 * sx)) & 3) : (wm * 2 + 0) * 9 + tap;
            const int peR = PROT ? (wm * 3 + r) * 6 + 2 + ((0x2c >> (2 * sx)) & 3) : (wm * 2 + 1) * 9 + tap;
            const char* pL = sPatch + ((chunk & 1) * H3_PATCH_ROWS + peL) * 128;
            const char* pR = sPatch + ((chunk & 1) * H3_PATCH_ROWS + peR) * 128;
            const int keyL = PROT ? ((0x90 >> (3 * sx)) & 7) : (peL >> 1) & 7, keyR = PROT ? ((0x36 >> (3 * sx)) & 7) : (peR >> 1) & 7;
            const bool isL = hasL && frow == 1, isR = hasR && frow == 14;
            auto ldP = [&](int j, int kk) -> u32x4 {
                const char* p = bP + j * 16 * BKB + ((H3_ROT ? ((fq + 4 * kk + fswP) & 7) : ((fq + 4 * kk) ^ fswP)) << 4);
                if (FOLD) {
                    if (j == 0 && isL) p = pL + ((PROT ? ((fq + 4 * kk + keyL) & 7) : ((fq + 4 * kk) ^ keyL)) << 4);
                    if (j == MI - 1 && isR) p = pR + ((PROT ? ((fq + 4 * kk + keyR) & 7) : ((fq + 4 * kk) ^ keyR)) << 4);
                }
                return ld16(p);
            };
            if (abl != 2) {
              if constexpr (!LAG) {
                // Fragment pipeline: the reads are issued in the order the MFMA groups consume them -- k-step 0's weight fragments and the
                // first GJ pixel fragments, then (while group 0 multiplies) the next pixel group and k-step 1's weight fragments -- so the
                // first MFMA waits for 4 + GJ reads, not for the whole burst, and every later group finds its fragments landed.
                // (This needs hipcc to COUNT its waits, which it only does with the DMA and the waits in the forms above.)
                constexpr int GJ = H3_GJ, GPK = MI / GJ, NG = KK * GPK;
                u32x4 fw[KK][NI];
    #pragma unroll
                for (int i = 0; i < NI; ++i) fw[0][i] = ld16(bQ + i * 16 * BKB + (((fq + 0) ^ fswQ) << 4));
                u32x4 fp[2][GJ];
    #pragma unroll
                for (int jj = 0; jj < GJ; ++jj) fp[0][jj] = ldP(jj, 0);
                if constexpr (EARLY_FRAGS) {
                    __builtin_amdgcn_sched_barrier(0);           // (the eight reads above go out first)
                    issue_dmas();
                    __builtin_amdgcn_sched_barrier(0);
                }
    #pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int kk = g / GPK, jb = (g % GPK) * GJ;
                    if (g + 1 < NG) {
                        const int kn = (g + 1) / GPK, jn = ((g + 1) % GPK) * GJ;
    #pragma unroll
                        for (int jj = 0; jj < GJ; ++jj) fp[(g + 1) & 1][jj] = ldP(jn + jj, kn);
                    }
                    if (g == 0) {
    #pragma unroll
                        for (int i = 0; i < NI; ++i) fw[1][i] = ld16(bQ + i * 16 * BKB + (((fq + 4) ^ fswQ) << 4));
                    }
                    __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                    for (int jj = 0; jj < GJ; ++jj)
    #pragma unroll
                        for (int i = 0; i < NI; ++i)
                            acc[i][jb + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, fw[kk][i]), __builtin_bit_cast(bf16x8, fp[g & 1][jj]), acc[i][jb + jj], 0, 0, 0);
                    if constexpr (NORM) {
                        // Rows land one step after their DMA was issued (the barrier below) and are first read two or more steps later
                        // (see the refill schedule above): the step in between is where they are normalised, by the waves without
                        // DMAs.  WHERE in the step: those waves have the matrix pipe to themselves while their SIMD partners issue the
                        // DMAs (about half of the step's 64 MFMAs), then share it.  The ~1 300 cycles of VALU / LDS work of a row go
                        // in the MIDDLE -- the partner wave, on the critical path, takes the whole pipe meanwhile; after the last
                        // MFMA the same work measured fully exposed (paired launch +20 us).
                        if (!(H3_NORM_ABL & 2) && g == H3_NORM_AT && (wave >> 2) != H3_HALO_HALF) {
                            __builtin_amdgcn_sched_barrier(0);
                            const int vw = wave & 3;
                            if (tap == 1 && chunk > 0) norm_row(2, chunk, vw, 4);
                            if (tap == 4 && chunk > 0) norm_row(3, chunk, vw, 4);
                            if (tap == 5 && chunk + 1 < nchunk) norm_row(0, chunk + 1, vw, 4);
                            if (tap == 7 && chunk + 1 < nchunk) norm_row(1, chunk + 1, vw, 4);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
              } else {
                // lagging half: k-step 0 now (two groups of GJ pixel fragments), k-step 1's fragments only LOADED -- into registers the
                // k-step 0 fragments have vacated -- and multiplied after the barrier (above / the tail behind the loop)
                static_assert(MI == 2 * H3_GJ && KK == 2, "lagging pipeline: two pixel groups, two k-steps");
                constexpr int GJ = H3_GJ;
                u32x4 fw0[NI], fp[2][GJ];
#pragma unroll
                for (int i = 0; i < NI; ++i) fw0[i] = ld16(bQ + i * 16 * BKB + (((fq + 0) ^ fswQ) << 4));
#pragma unroll
                for (int jj = 0; jj < GJ; ++jj) fp[0][jj] = ldP(jj, 0);
#pragma unroll
                for (int jj = 0; jj < GJ; ++jj) fp[1][jj] = ldP(GJ + jj, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jj = 0; jj < GJ; ++jj)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw0[i]), __builtin_bit_cast(bf16x8, fp[0][jj]), acc[i][jj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);           // (the loads below reuse the first group's registers: not above its MFMAs)
#pragma unroll
                for (int i = 0; i < NI; ++i) cw[i] = ld16(bQ + i * 16 * BKB + (((fq + 4) ^ fswQ) << 4));
#pragma unroll
                for (int jj = 0; jj < GJ; ++jj) cp[jj] = ldP(jj, 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jj = 0; jj < GJ; ++jj)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
                        acc[i][GJ + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw0[i]), __builtin_bit_cast(bf16x8, fp[1][jj]), acc[i][GJ + jj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jj = 0; jj < GJ; ++jj) cp[GJ + jj] = ldP(GJ + jj, 1);
              }
            }
            H3_STAMP(2);
            if (LATE_HALO && (wave >> 2) != H3_ISSUER_HALF) {
                // this tile's pieces stay in flight; everything older (the pieces of the tile before) must have landed
                int mine = 0;
                if (abl == 0 || abl == 2) {
                    const int vw = wave & 3;
                    const bool patch = FOLD && tap == 3 && chunk + 1 < nchunk;
                    const int rows = issue_halo(vw);
                    mine = rows * (vw == 0 ? 5 : 4) + (patch ? (vw == 0 ? 2 : 1) : 0);
                }
                switch (mine) {
                    case 0: sgg_wait_vm<0>(); break;
                    case 4: sgg_wait_vm<4>(); break;
                    case 5: sgg_wait_vm<5>(); break;
                    case 7: sgg_wait_vm<7>(); break;
                    case 8: sgg_wait_vm<8>(); break;
                    case 9: sgg_wait_vm<9>(); break;
                    case 10: sgg_wait_vm<10>(); break;
                    case 12: sgg_wait_vm<12>(); break;
                    default: sgg_wait_vm<0>(); break;
                }
            } else
            if (!NORM || (wave >> 2) == H3_HALO_HALF || (wave >> 2) == H3_ISSUER_HALF) SGG_WAIT_VM0();   // (NORM: the other waves only have stores in flight)
            H3_STAMP(3);
            SGG_WAIT_LGKM0();
            H3_STAMP(4);
            if (abl < 3) __builtin_amdgcn_s_barrier();
            H3_STAMP(5);
            if (++tap == 9) { tap = 0; ++chunk; }
        }
        if constexpr (LAG) {
            if (ntiles > 0 && abl != 2) {
#pragma unroll
                for (int j = 0; j < MI; ++j)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cw[i]), __builtin_bit_cast(bf16x8, cp[j]), acc[i][j], 0, 0, 0);
            }
        }
    };
    if (STAG && (wave >> 2) == H3_LAG_HALF) main_loop(std::true_type{}); else main_loop(std::false_type{});

    if (dbg_clk) { g_dbg_clk[2] = clock64(); g_dbg_clk[3] = wall_clock64(); }
    if (abl == 6) return;
    // the next tile's prologue (PERSIST): every wave past the main loop's last barrier, so the halo, patch buffer 0 and weight
    // stage 0 are free.  Called once per tile from the epilogue, behind its loads and in front of its stores.
    auto prime_next = [&]() {
        primed = false;
        if constexpr (PERSIST) {
            if (it + 1 < tpb && tile_id + 1 < total_tiles && abl == 0) {
                int lop = lane, nid = tile_id + 1;
                asm volatile("" : "+v"(lop), "+s"(nid));             // (nothing of the next tile's addresses is computed ahead of this point)
                const Tile nx = tile_of(nid);
#pragma unroll
                for (int k = 0; k < 4; ++k) load_halo_row(nx, k, 0, wave, 8, lop);
                load_patch(nx, 0, wave, 8, lop);
                load_w(nx, 0, 0, MODE == MODE_FWD ? 0 : 8, wave, 8, lop);
                primed = true;
            }
        }
    };
    // Epilogue, one 16-channel group at a time.  STATS 1 (forward): per-channel (sum, sumsq) of the STORED output for the
    // instance norm that follows the conv.  STATS 2 (data gradient): the first pass of the instance-norm BACKWARD that
    // consumes this gradient, (sum g, sum g*xhat) with g = dx * act'(gamma*xhat + beta) and xhat from that norm's input
    // a.nx and statistics a.nstats (norm.hip in_partial_kernel<BWD>).  Each wave reduces its 128 pixels x 64 channels
    // (registers -> 16-lane butterfly, fixed order) and writes one row per channel as pixel chunk (tile, wave row) of the
    // image, so the norm skips its own statistics pass over the tensor.
    const size_t prow = ((size_t)img * a.H + h0 + wm) * a.W + w0;
    if constexpr (PERSIST && H3_PRIME_FIRST) prime_next();
    if constexpr (STATS != 2) {
        // pixel-major store order: the four 16-channel groups of a pixel go out back to back, so the 128 bytes a wave
        // writes per pixel merge into whole lines in L2 (channel-group-major order cost +47 MB of HBM fetches per launch:
        // partially written lines are read back)
        const float* const bias_n = PAIR && img >= a.nsplit ? a.bias2 : a.bias;   // (picked here, not before the loop: registers)
        float bv[NI][4], s1[NI][4], s2[NI][4];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int dc = n0 + wn * WN + i * 16 + fq * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[i][e] = (bias_n && dc < DC) ? bias_n[dc + e] : 0.f; s1[i][e] = s2[i][e] = 0.f; }
        }
        // PLAIN path (every call of the step: no activation, bf16 result, bf16 or no addend): the uniform conditions are tested
        // once, and the skip-gradient addend's 32 loads per lane are all issued before the first store -- the fragment registers
        // are dead here.  (With the tests inside the unrolled loops every load was followed by s_waitcnt vmcnt(0): 32 dependent
        // round trips at the tail of a grid that has nothing else to overlap them with.)
        const bool plain = a.act == SGG_ACT_NONE && !(MODE != MODE_FWD && (a.dst_f32 || a.addend_f32));
        if (plain && H3_WIDE) {
            // A lane holds 4 consecutive channels of one pixel per fragment: 8-byte stores, 32 per lane, and the tail of a grid
            // whose blocks all store at once is store-ISSUE bound.  Fragments 2 jp and 2 jp + 1 trade their halves between lane
            // rows (row_swap16) so that every lane owns 8 consecutive channels of ONE pixel -- lanes of rows 0 / 2 a pixel of the
            // even fragment, rows 1 / 3 one of the odd fragment: 16 stores of 16 bytes (and 16 addend loads instead of 32).  Same
            // values, same roundings, same statistics (taken before the exchange); pixel-major order as before.
            auto run = [&](auto has_add) {
                constexpr bool ADD = decltype(has_add)::value;
                const int jo = fq & 1, cb = (fq >> 1) * 8;
                const size_t e0 = (prow + jo * 16 + frow) * DC + n0 + wn * WN + cb;   // element (fragment pair 0, group 0) of this lane
                const size_t ej2 = (size_t)32 * DC;
                // (persistent form: the addend in two halves of 8 loads -- with all 16 in flight next to the accumulators and the
                // tile loop's state the data gradient spilled 30-130 registers)
                constexpr int NH = (ADD && PERSIST) ? 2 : 1, JH = MI / 2 / NH;
                if constexpr (!(PERSIST && H3_PRIME_FIRST)) { if constexpr (!ADD) prime_next(); }
#pragma unroll
                for (int hb = 0; hb < NH; ++hb) {
                    u32x4 adv[JH][NI];
                    if constexpr (ADD) {
#pragma unroll
                        for (int jq = 0; jq < JH; ++jq)
#pragma unroll
                            for (int i = 0; i < NI; ++i) {
                                adv[jq][i] = zero16();
                                if (n0 + wn * WN + i * 16 + cb < DC) {
                                    const bf16* ap = reinterpret_cast<const bf16*>(a.addend) + e0 + (hb * JH + jq) * ej2 + i * 16;
                                    adv[jq][i] = SGG_NT_ADDEND ? ld16_nt(ap) : ld16(ap);
                                }
                            }
                        if constexpr (!(PERSIST && H3_PRIME_FIRST)) { if (hb == 0) prime_next(); }
                    }
#pragma unroll
                    for (int jq = 0; jq < JH; ++jq) {
                        const int jp = hb * JH + jq;
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            float v0[4], v1[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v0[e] = acc[i][2 * jp][e] + bv[i][e]; v1[e] = acc[i][2 * jp + 1][e] + bv[i][e]; }
                            u32x4 pk;
                            if constexpr (ADD) {
                                float ad[8], o[8];
                                ET<bf16>::unpack(adv[jq][i], ad);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { row_swap16(v0[e], v1[e]); o[e] = v0[e] + ad[e]; o[4 + e] = v1[e] + ad[4 + e]; }
                                pk = ET<bf16>::pack(o);
                            } else {
                                const bf16x4 p0 = {(bf16)v0[0], (bf16)v0[1], (bf16)v0[2], (bf16)v0[3]}, p1 = {(bf16)v1[0], (bf16)v1[1], (bf16)v1[2], (bf16)v1[3]};
                                if (STATS == 1) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {           // (fragment order as in the 8-byte form: 2 jp, then 2 jp + 1)
                                        const float r0 = (float)p0[e]; s1[i][e] += r0; s2[i][e] += r0 * r0;
                                    }
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float r1 = (float)p1[e]; s1[i][e] += r1; s2[i][e] += r1 * r1;
                                    }
                                }
                                const u32x2 q0 = __builtin_bit_cast(u32x2, p0), q1 = __builtin_bit_cast(u32x2, p1);
                                uint32_t a0 = q0[0], a1 = q0[1], b0 = q1[0], b1 = q1[1];
                                row_swap16(a0, b0); row_swap16(a1, b1);
                                pk = (u32x4){a0, a1, b0, b1};
                            }
                            if (n0 + wn * WN + i * 16 + cb < DC) st16(reinterpret_cast<bf16*>(a.dst) + e0 + jp * ej2 + i * 16, pk);
                        }
                    }
                }
            };
            if (MODE != MODE_FWD && a.addend) run(std::true_type{}); else run(std::false_type{});
        } else if (plain) {
            auto run = [&](auto has_add) {
                constexpr bool ADD = decltype(has_add)::value;
                const size_t e0 = (prow + frow) * DC + n0 + wn * WN + fq * 4;      // element (pixel j = 0, group i = 0) of this lane
                const size_t ej = (size_t)16 * DC;
                bf16x4 adv[MI][NI];
                if constexpr (ADD) {
#pragma unroll
                    for (int j = 0; j < MI; ++j)
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            const int dc = n0 + wn * WN + i * 16 + fq * 4;
                            adv[j][i] = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
                            if (dc < DC) adv[j][i] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(a.addend) + e0 + j * ej + i * 16);
                        }
                }
                if constexpr (!(PERSIST && H3_PRIME_FIRST)) prime_next();
#pragma unroll
                for (int j = 0; j < MI; ++j) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int dc = n0 + wn * WN + i * 16 + fq * 4;
                        if (dc >= DC) continue;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + bv[i][e];
                        if constexpr (ADD) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)adv[j][i][e];
                        }
                        bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dst) + e0 + j * ej + i * 16) = pk;
                        if (STATS == 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { const float vr = (float)pk[e]; s1[i][e] += vr; s2[i][e] += vr * vr; }
                        }
                    }
                }
            };
            if (MODE != MODE_FWD && a.addend) run(std::true_type{}); else run(std::false_type{});
        } else {
            if constexpr (!(PERSIST && H3_PRIME_FIRST)) prime_next();
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const size_t dpix = prow + j * 16 + frow;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int dc = n0 + wn * WN + i * 16 + fq * 4;
                    if (dc >= DC) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_apply(acc[i][j][e] + bv[i][e], a.act, a.leak);
                    if (a.addend) {
                        if (MODE != MODE_FWD && a.addend_f32) {
                            const f32x4 ad = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.addend) + dpix * DC + dc);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += ad[e];
                        } else {
                            const bf16* ad = reinterpret_cast<const bf16*>(a.addend) + dpix * DC + dc;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)ad[e];
                        }
                    }
                    if (MODE != MODE_FWD && a.dst_f32) {      // mixed mode: the data gradient stays f32 on its way to the next norm backward
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.dst) + dpix * DC + dc) = (f32x4){v[0], v[1], v[2], v[3]};
                        continue;
                    }
                    bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dst) + dpix * DC + dc) = pk;
                    if (STATS == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float vr = (float)pk[e]; s1[i][e] += vr; s2[i][e] += vr * vr; }
                    }
                }
            }
        }
        if (STATS == 1) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) {
                        s1[i][e] += __shfl_xor(s1[i][e], off);
                        s2[i][e] += __shfl_xor(s2[i][e], off);
                    }
                }
            if (frow == 0) {
                const int chunks = tilesH * tilesW * 2;
                const int chunk = (th * tilesW + tw) * 2 + wm;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int dc = n0 + wn * WN + i * 16 + fq * 4;
                    if (dc >= DC) continue;
                    float* o = a.stats + (((size_t)img * chunks + chunk) * DC + dc) * 2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[2 * e] = s1[i][e]; o[2 * e + 1] = s2[i][e]; }
                }
            }
        }
    }
    // STATS 2 (data gradient; no bias, no activation -- run_gemm checks).  The norm's input tile (2 rows x 128 pixels x 256
    // channels = 128 KB) is first DMA'd into LDS -- the halo and the weight stages are free now -- with coalesced 16-byte
    // pieces, and each lane then picks its 8-byte (pixel, 4 channels) groups out of LDS: 512-byte pixel rows, 16-byte chunk c
    // of pixel p at position c ^ (p & 31), so the 16 pixels a ds_read_b64 touches hit 16 different chunks.
    // Per lane and channel only two constants stay in registers, the norm's affine form A = gamma*rstd, B = beta - mean*A
    // (the sign of A*x + B is the activation mask, as in in_apply_kernel); the sums are (sum g, sum g*x) and the chunk's
    // second sum is recovered at the end as rstd * (sum g*x - mean * sum g).  Pixel-major store order as above; two pixels'
    // addend loads and LDS reads are in flight at a time.
    if constexpr (STATS == 2) {
        for (int id = wave; id < 128; id += 8) {         // 128 wave-instructions of 2 pixels x 512 B
            const int p = id * 2 + (lane >> 5), pos = lane & 31;
            const int chunk = pos ^ (p & 31);
            const bool ok = n0 + chunk * 8 < DC;
            const char* src = ok ? a.nx + ((((size_t)img * a.H + h0 + (p >> 7)) * a.W + w0 + (p & 127)) * DC + n0 + chunk * 8) * 2 : zero;
            dma16_to_lds(src, (__attribute__((address_space(3))) void*)(lH + id * 1024));
        }
        float cA[NI][4], cB[NI][4], s1[NI][4], s2[NI][4];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int dc = n0 + wn * WN + i * 16 + fq * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cA[i][e] = cB[i][e] = s1[i][e] = s2[i][e] = 0.f;
                if (dc < DC) {
                    const float mu = a.nstats[((size_t)img * DC + dc + e) * 2], rs = a.nstats[((size_t)img * DC + dc + e) * 2 + 1];
                    cA[i][e] = a.ngamma[dc + e] * rs;
                    cB[i][e] = a.nbeta[dc + e] - mu * cA[i][e];
                }
            }
        }
        SGG_WAIT_VM0();
        __builtin_amdgcn_s_barrier();
        const size_t e0 = (prow + frow) * DC + n0 + wn * WN + fq * 4;
        const size_t ej = (size_t)16 * DC;
        act_dispatch(a.nact, [&](auto nact_c) {
            constexpr int NACT = decltype(nact_c)::value;
            auto run = [&](auto has_add) {
                constexpr bool ADD = decltype(has_add)::value;
#pragma unroll
                for (int j0 = 0; j0 < MI; j0 += 2) {
                    bf16x4 adv[2][NI], xq[2][NI];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            const int j = j0 + jj, dc = n0 + wn * WN + i * 16 + fq * 4;
                            const int pl = wm * 128 + j * 16 + frow, cl = wn * WN + i * 16 + fq * 4;      // tile pixel, tile channel
                            xq[jj][i] = *reinterpret_cast<const bf16x4*>(smem + (size_t)pl * 512 + (((cl >> 3) ^ (pl & 31)) << 4) + (cl & 7) * 2);
                            if constexpr (ADD) {
                                adv[jj][i] = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
                                if (dc < DC) adv[jj][i] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(a.addend) + e0 + j * ej + i * 16);
                            }
                        }
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            const int j = j0 + jj, dc = n0 + wn * WN + i * 16 + fq * 4;
                            if (dc >= DC) continue;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
                            if constexpr (ADD) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += (float)adv[jj][i][e];
                            }
                            bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dst) + e0 + j * ej + i * 16) = pk;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float xv = (float)xq[jj][i][e];
                                const float g = (float)pk[e] * act_grad_c<NACT>(cA[i][e] * xv + cB[i][e], a.nleak);
                                s1[i][e] += g; s2[i][e] += g * xv;
                            }
                        }
                }
            };
            if (a.addend) run(std::true_type{}); else run(std::false_type{});
        });
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    s1[i][e] += __shfl_xor(s1[i][e], off);
                    s2[i][e] += __shfl_xor(s2[i][e], off);
                }
            }
        if (frow == 0) {
            const int chunks = tilesH * tilesW * 2;
            const int chunk = (th * tilesW + tw) * 2 + wm;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int dc = n0 + wn * WN + i * 16 + fq * 4;
                if (dc >= DC) continue;
                float* o = a.stats + (((size_t)img * chunks + chunk) * DC + dc) * 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float mu = a.nstats[((size_t)img * DC + dc + e) * 2], rs = a.nstats[((size_t)img * DC + dc + e) * 2 + 1];
                    o[2 * e] = s1[i][e]; o[2 * e + 1] = rs * (s2[i][e] - mu * s1[i][e]);
                }
            }
        }
    }
  }   // tile loop
}

// -------------------------------------------------------------------------------------------------
// 3x3 STRIDE-2 data gradient / Conv2DTranspose forward with the dy halo resident in LDS and all four output-parity
// classes computed by one block (module.py:230-236 backward, :254-258 forward; D's module.py:284-294 backward).
//
// The generic kernel gives every parity class its own blocks with 128x64 tiles: K loops of 2-8 tiles, 43 FLOP per staged
// byte, and every dy pixel fetched once per tap.  Here a block owns an 8 x 32 tile of dy pixels = a 16 x 64 tile of
// output pixels x 64 output channels: per 64-channel chunk of dy the 10 x 34 halo is staged once (double-buffered
// across chunks) and the nine taps -- 4 + 2 + 2 + 1 over the classes -- read shifted rows of it; weights stream per
// kernel row (3 taps, 24 KB) through a 2-stage ring.  3.6x fewer L2->LDS bytes, no per-class prologue.
// Wave w owns dy row w of the tile (two 16-pixel fragments) x all 64 channels x 4 classes = 32 MFMA tiles.
// Output pixel (2i+a, 2j+b) of class (a,b) takes tap (r,s) with a = (r+pad_t)&1, b = (s+pad_l)&1 from dy pixel
// (i + (a+pad_t-r)/2, j + (b+pad_l-s)/2), zero outside dy.
// -------------------------------------------------------------------------------------------------
#ifndef S2_WIDE
#define S2_WIDE 1                                      // epilogue with 16-byte stores (two fragments exchanged between lane rows)
#endif
#define S2_TI 8
#define S2_TJ 32
#define S2_PITCH 40                                    // halo row pitch in pixels (34 used)
#define S2_HALO_BYTES ((S2_TI + 2) * S2_PITCH * 128)   // 51200
#define S2_WSTAGE (3 * 64 * 128)                       // one kernel row of taps x 64 output channels x 128 B
#define S2_STATS_BYTES (8 * 64 * 2 * 4)                // per-wave (sum, sumsq) rows of a tile's 64 channels (forward with the norm-statistics epilogue)
#define S2_LDS (2 * S2_HALO_BYTES + 2 * S2_WSTAGE + S2_STATS_BYTES)     // 155648

// STATS: the Conv2DTranspose-forward build with the norm-statistics epilogue -- its own instantiation, because the kernel sits at 251 VGPRs and
// the 32 running sums tipped the common build into 192 spilled registers (-4.5 % on the whole step) when they shared it.
template <int PT, int PL, bool STATS = false>          // pad_t, pad_l (0 or 1): compile-time so that each tap's class is static
__global__ __launch_bounds__(512) void deconv_s2_halo_kernel(ConvArgs a, int total_tiles, int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sH = smem;                                   // 2 x halo [10][40][128 B]
    char* sW = smem + 2 * S2_HALO_BYTES;               // 2 x [3 taps][64 c][128 B]
    float* sS = reinterpret_cast<float*>(smem + 2 * S2_HALO_BYTES + 2 * S2_WSTAGE);   // [8 waves][64 channels][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int SC = a.K, DC = a.C;                      // dy channels (reduction), dx channels
    const int wrow = 9 * SC, nchunk = SC >> 6;
    const int tilesN = DC >> 6, tilesJ = a.Wo / S2_TJ, tilesI = a.Ho / S2_TI;
    // Persistent blocks: block `lid` (XCD-contiguous numbering) walks tiles lid*tiles_per_block ... in order, so the
    // prologue of a tile (halo + first weights) is fetched under the last stage of the previous one and the output
    // stores of a tile drain under the next tile's MFMAs; consecutive tiles share halo columns (L2 hits).
    int lid;
    {
        const int b = (int)blockIdx.x, nm = (int)gridDim.x;
        const int q = nm >> 3, rr = nm & 7, xcd = b & 7;
        lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
    }
    const int t_beg = lid * tiles_per_block;
    const int t_end = min(total_tiles, t_beg + tiles_per_block);
    if (t_beg >= t_end) return;
    struct Tile { int img, i0, j0, n0; };
    auto decode = [&](int t) {
        Tile T;
        T.n0 = (t % tilesN) * 64;
        int mt = t / tilesN;
        T.j0 = (mt % tilesJ) * S2_TJ; mt /= tilesJ;
        T.i0 = (mt % tilesI) * S2_TI;
        T.img = mt / tilesI;
        return T;
    };
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    const int hpos = lane & 7, hsub = lane >> 3;

    // halo of one 64-channel chunk: 10 rows x 5 DMAs of 8 pixels; wave w issues q = w, w+8, ...
    auto load_halo = [&](int buf, const Tile& T, int chunk) {
        char* dstb = sH + buf * S2_HALO_BYTES;
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int q = wave + 8 * it;
            if (q >= (S2_TI + 2) * 5) break;
            const int k = q / 5, cg = q - 5 * k;
            const int hp = cg * 8 + hsub;
            const int hi = T.i0 - 1 + k, wi = T.j0 - 1 + hp;
            const bool ok = hp < S2_TJ + 2 && (unsigned)hi < (unsigned)a.Ho && (unsigned)wi < (unsigned)a.Wo;
            // rotation swizzle of the halo rows (conflict-free fragment reads for every tap shift; see conv3x3_halo_gemm_kernel)
            const int hrow_ = k * S2_PITCH + hp;
            const int schunk = H3_ROT ? ((hpos - (hrow_ & 6)) & 7) : (hpos ^ ((hrow_ >> 1) & 7));
            const char* src = ok ? a.src + ((((size_t)T.img * a.Ho + hi) * a.Wo + wi) * SC + chunk * 64) * 2 + (schunk << 4) : zero;
            dma16_to_lds(src, (__attribute__((address_space(3))) void*)(dstb + (k * S2_PITCH + cg * 8) * 128));
        }
    };
    // weights of kernel row r, chunk: 3 taps x 64 rows x 128 B = 24 DMAs of 8 rows; wave w issues 3w..3w+2
    auto load_w = [&](int stg, const Tile& T, int chunk, int r) {
        char* dstb = sW + stg * S2_WSTAGE;
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int q = wave * 3 + it;                // 0..23: tap column s = q / 8, rows 8*(q%8)..
            const int sx = q >> 3, row = (q & 7) * 8 + hsub;
            const int key = (((sx * 64 + row) >> 1) & 7);
            const char* wm = T.img >= a.nsplit ? a.wmat2 : a.wmat;      // stacked batch of two networks: per-image weights
            const char* src = wm + ((size_t)(T.n0 + row) * wrow + (size_t)(r * 3 + sx) * SC + chunk * 64) * 2 + ((hpos ^ key) << 4);
            dma16_to_lds(src, (__attribute__((address_space(3))) void*)(dstb + q * 1024));
        }
    };

    f32x4 acc[4][2][4];                                // [class][pixel fragment][channel tile]
    const int frow = lane & 15, fq = lane >> 4;
    Tile cur = decode(t_beg);
    load_halo(0, cur, 0);
    load_w(0, cur, 0, 0);
    SGG_WAIT_VM0();
    __builtin_amdgcn_s_barrier();

    const int nqc = (t_end - t_beg) * nchunk;          // (tile, chunk) pairs of this block
    int chunk = 0, tcur = t_beg;
    for (int qc = 0; qc < nqc; ++qc) {
        if (chunk == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[c][j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // the (tile, chunk) after this one
        const bool more = qc + 1 < nqc;
        const bool last_chunk = chunk + 1 == nchunk;
        const Tile nxt = (more && last_chunk) ? decode(tcur + 1) : cur;
        const int nchk = last_chunk ? 0 : chunk + 1;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int st = qc * 3 + r;
            if (r < 2) load_w((st + 1) & 1, cur, chunk, r + 1);
            else if (more) load_w((st + 1) & 1, nxt, nchk, 0);
            if (r == 0 && more) load_halo((qc + 1) & 1, nxt, nchk);     // that buffer was last read in pair qc-1
            const int ar = (r + PT) & 1, dr = (ar + PT - r) / 2;          // class row parity, dy row offset (compile time)
            const char* hb = sH + (qc & 1) * S2_HALO_BYTES;
            const char* wb = sW + (st & 1) * S2_WSTAGE;
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                const int bs = (sx + PL) & 1, ds = (bs + PL - sx) / 2;
                const int cls = ar * 2 + bs;
                const int hrow = (wave + dr + 1) * S2_PITCH + (ds + 1) + frow;   // halo pixel of fragment 0, this lane
                const int fswP = H3_ROT ? (hrow & 6) : ((hrow >> 1) & 7);
                const char* bP = hb + hrow * 128;
                const int wr = sx * 64 + frow;
                const int fswQ = (wr >> 1) & 7;
                const char* bQ = wb + wr * 128;
                u32x4 fw[2][4], fp[2][2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) fw[kk][i] = ld16(bQ + i * 16 * 128 + (((fq + 4 * kk) ^ fswQ) << 4));
#pragma unroll
                    for (int j = 0; j < 2; ++j) fp[kk][j] = ld16(bP + j * 16 * 128 + ((H3_ROT ? ((fq + 4 * kk + fswP) & 7) : ((fq + 4 * kk) ^ fswP)) << 4));
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[cls][j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[kk][i]),
                                                                                      __builtin_bit_cast(bf16x8, fp[kk][j]), acc[cls][j][i], 0, 0, 0);
            }
            SGG_WAIT_VM0();
            SGG_WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();
        }
        if (last_chunk) {
            // epilogue of tile `cur`; the stores drain while the next tile is multiplied
            float bv[4][4];
            const float* const bias_t = cur.img >= a.nsplit ? a.bias2 : a.bias;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[i][e] = bias_t ? bias_t[cur.n0 + i * 16 + fq * 4 + e] : 0.f;
            auto epilogue = [&](auto act_c) {
                constexpr int ACT = decltype(act_c)::value;
                auto store_tile = [&](auto add_c, auto st_c) {
                    constexpr bool ADD = decltype(add_c)::value;
                    constexpr bool ST = STATS && decltype(st_c)::value && !ADD && S2_WIDE;      // Conv2DTranspose forward + the following norm's sums
                    if constexpr (S2_WIDE) {
                        // 16-byte stores: the two pixel fragments of a class trade halves between lane rows (row_swap16, see the
                        // 3x3 halo GEMM's epilogue) -- 16 store instructions per wave and tile instead of 32; the layers this
                        // kernel serves move more output bytes per FLOP than any other GEMM of the step
                        const int jo = fq & 1, cb = (fq >> 1) * 8;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int h = 2 * (cur.i0 + wave) + (c >> 1);
                            const int w = 2 * (cur.j0 + jo * 16 + frow) + (c & 1);
                            const size_t dpix = ((size_t)cur.img * a.H + h) * a.W + w;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int dc = cur.n0 + i * 16 + cb;
                                float v0[4], v1[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    v0[e] = act_apply_c<ACT>(acc[c][0][i][e] + bv[i][e], a.leak);
                                    v1[e] = act_apply_c<ACT>(acc[c][1][i][e] + bv[i][e], a.leak);
                                }
                                u32x4 pk;
                                if constexpr (ADD) {
                                    float ad[8], o[8];
                                    ET<bf16>::unpack(ld16(reinterpret_cast<const bf16*>(a.addend) + dpix * DC + dc), ad);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { row_swap16(v0[e], v1[e]); o[e] = v0[e] + ad[e]; o[4 + e] = v1[e] + ad[4 + e]; }
                                    pk = ET<bf16>::pack(o);
                                } else {
                                    const bf16x4 p0 = {(bf16)v0[0], (bf16)v0[1], (bf16)v0[2], (bf16)v0[3]}, p1 = {(bf16)v1[0], (bf16)v1[1], (bf16)v1[2], (bf16)v1[3]};
                                    const u32x2 q0 = __builtin_bit_cast(u32x2, p0), q1 = __builtin_bit_cast(u32x2, p1);
                                    uint32_t a0 = q0[0], a1 = q0[1], b0 = q1[0], b1 = q1[1];
                                    row_swap16(a0, b0); row_swap16(a1, b1);
                                    pk = (u32x4){a0, a1, b0, b1};
                                }
                                st16(reinterpret_cast<bf16*>(a.dst) + dpix * DC + dc, pk);
                            }
                        }
                        if constexpr (ST) {
                            // Statistics of the STORED values (channel 16 i + 4 fq + e of this lane), as a SECOND pass over the accumulators,
                            // one channel group at a time: 8 running sums instead of 32 next to 128 accumulators (the kernel has no
                            // registers to spare; the re-evaluation is a few hundred VALU instructions per tile).  Wave: 16-lane butterfly
                            // (fixed order) -> one (sum, sumsq) per channel; block: the eight waves' rows through LDS, summed in wave
                            // order by 128 threads -> ONE partial row per (image, pixel tile), channels n0 .. n0 + 63
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int c = 0; c < 4; ++c)
#pragma unroll
                                    for (int j = 0; j < 2; ++j)
#pragma unroll
                                        for (int e = 0; e < 4; ++e) {
                                            const float r = (float)(bf16)act_apply_c<ACT>(acc[c][j][i][e] + bv[i][e], a.leak);
                                            s1[e] += r; s2[e] += r * r;
                                        }
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
#pragma unroll
                                    for (int off = 1; off < 16; off <<= 1) {
                                        s1[e] += __shfl_xor(s1[e], off);
                                        s2[e] += __shfl_xor(s2[e], off);
                                    }
                                }
                                if (frow == 0) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        float* o = sS + ((wave * 64) + i * 16 + fq * 4 + e) * 2;
                                        o[0] = s1[e]; o[1] = s2[e];
                                    }
                                }
                            }
                            SGG_WAIT_LGKM0();
                            __builtin_amdgcn_s_barrier();
                            if (tid < 128) {
                                const int ch = tid >> 1, which = tid & 1;
                                float t = 0.f;
#pragma unroll
                                for (int w8 = 0; w8 < 8; ++w8) t += sS[(w8 * 64 + ch) * 2 + which];
                                const int chunks = tilesI * tilesJ;
                                const int chunk = (cur.i0 / S2_TI) * tilesJ + cur.j0 / S2_TJ;
                                a.stats[(((size_t)cur.img * chunks + chunk) * DC + cur.n0 + ch) * 2 + which] = t;
                            }
                            // (the next tile's rows are written at least a main-loop barrier later)
                        }
                        return;
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int h = 2 * (cur.i0 + wave) + (c >> 1);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int w = 2 * (cur.j0 + j * 16 + frow) + (c & 1);
                            const size_t dpix = ((size_t)cur.img * a.H + h) * a.W + w;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int dc = cur.n0 + i * 16 + fq * 4;
                                float v[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = act_apply_c<ACT>(acc[c][j][i][e] + bv[i][e], a.leak);
                                if constexpr (ADD) {
                                    const bf16* ad = reinterpret_cast<const bf16*>(a.addend) + dpix * DC + dc;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] += (float)ad[e];
                                }
                                bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dst) + dpix * DC + dc) = pk;
                            }
                        }
                    }
                };
                if constexpr (STATS) store_tile(std::false_type{}, std::true_type{});
                else if (a.addend) store_tile(std::true_type{}, std::false_type{});
                else store_tile(std::false_type{}, std::false_type{});
            };
            // (the statistics build serves a layer followed by a norm: no activation, and no tanh code next to its second pass)
            if constexpr (STATS) epilogue(std::integral_constant<int, SGG_ACT_NONE>{}); else act_dispatch(a.act, epilogue);
            cur = nxt; ++tcur; chunk = 0;
        } else ++chunk;
    }
}

static bool s2halo_ok(const ConvArgs& a, bool is_bf16) {
    const int en = sgg_config().s2halo;
    if (!en || !is_bf16 || !use_glds() || a.ksplit > 1 || SGG_ABLATE_OF(a) || a.reflect) return false;
    if (a.R != 3 || a.S != 3 || a.stride != 2 || a.H != 2 * a.Ho || a.W != 2 * a.Wo) return false;
    if (a.pad_t != 0 || a.pad_l != 0) return false;     // TF 'SAME' with an even input pads bottom/right only
    return a.K % 64 == 0 && a.C % 64 == 0 && a.Wo % S2_TJ == 0 && a.Ho % S2_TI == 0;
}

template <int PT, int PL, bool STATS = false>
static int launch_s2halo_p(const ConvArgs& a, hipStream_t s) {
    auto kern = deconv_s2_halo_kernel<PT, PL, STATS>;
    SGG_LDS_ATTR(kern, S2_LDS);
    const int total = (int)((int64_t)a.N * (a.Ho / S2_TI) * (a.Wo / S2_TJ) * (a.C / 64));
    const int tpb = (total + 255) / 256;               // one persistent block per CU
    const int blocks = (total + tpb - 1) / tpb;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), S2_LDS, s, a, total, tpb);
    return sgg_check_launch();
}
static int launch_s2halo(const ConvArgs& a, hipStream_t s) {
    if (a.stats) return a.addend ? SGG_EUNSUPPORTED : launch_s2halo_p<0, 0, true>(a, s);
    return launch_s2halo_p<0, 0>(a, s);
}

// shapes the halo-resident kernel takes: 3x3, stride 1, pad 1 on a same-size output, bf16, 64 | source channels,
// 128 | W, even H; REFLECT only forward (the REFLECT data gradient needs the mirrored border terms)
static bool halo3_ok(const ConvArgs& a, int mode, bool is_bf16) {
    const int en = sgg_config().halo3;
    if (!en || !is_bf16 || !use_glds() || a.ksplit > 1) return false;
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad_t != 1 || a.pad_l != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    if (a.W % H3_TW || (a.H & 1) || a.H < 4) return false;
    const int SC = mode == MODE_FWD ? a.C : a.K;
    if (SC % 64) return false;
    // buffer-descriptor DMAs: one image of the source and one tile's weight rows must stay below SGG_BUF_OOB bytes
    if ((int64_t)a.H * a.W * SC * 2 >= (int64_t)SGG_BUF_OOB || (int64_t)256 * 9 * SC * 2 >= (int64_t)SGG_BUF_OOB) return false;
    return mode == MODE_FWD || mode == MODE_DGRAD;
}

#define H3_NORM_MAXC 512                               // NORM: the A / B table behind the tiles holds this many source channels
template <int MODE, bool FOLD, int STATS = 0, bool PAIR = false, int SRC = 0>
static int launch_halo3(const ConvArgs& a, hipStream_t s) {
    auto kern = conv3x3_halo_gemm_kernel<MODE, FOLD, STATS, PAIR, SRC>;
    constexpr int lds = (FOLD ? H3_LDS_FOLD : H3_LDS) + (SRC == 2 ? H3_NORM_MAXC * 8 : 0);
    SGG_LDS_ATTR(kern, lds);
    const int DC = MODE == MODE_FWD ? a.K : a.C;
    int64_t blocks = (int64_t)a.N * (a.H / 2) * (a.W / H3_TW) * ((DC + 255) / 256);
    // persistent form: one block per CU (one fits: >= 134 KB of LDS), each walking ceil(tiles / blocks) consecutive tiles
    constexpr int ADDR = SRC == 0 ? H3_ADDR0 : (SRC == 1 ? H3_ADDR1 : H3_ADDR2);
    if (H3_PERSIST && SRC != 2 && ADDR == 0 && STATS != 2 && SGG_ABLATE_OF(a) == 0) {
        const int64_t cus = sgg_num_cus();
        if (blocks > cus) { const int64_t tpb = (blocks + cus - 1) / cus; blocks = (blocks + tpb - 1) / tpb; }
    }
    sgg_launch_timed(kern, dim3((unsigned)blocks), dim3(512), (unsigned)lds, s, a);
    return sgg_check_launch();
}

// Side tensor of the FOLD variant: [N][H][2 sides][9 taps][K] complete mirrored gather rows of the pixels in columns
// 1 and W-2, then [N][2][W][K] virtual rows dy[2]+dy[0] and dy[H-3]+dy[H-1]   (pad 1)
template <typename T>
__global__ __launch_bounds__(256) void fold_halo_gather_kernel(const char* dy, char* fold, int N, int H, int W, int K) {
    constexpr int VEC = ET<T>::VEC;
    const int cpv = K / VEC;
    const int64_t npatch = (int64_t)N * H * 18 * cpv, total = npatch + (int64_t)N * 2 * W * cpv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        if (i < npatch) {
            int ch = (int)(i % cpv);
            int64_t t = i / cpv;
            int tap = (int)(t % 9); t /= 9;
            int side = (int)(t & 1); t >>= 1;
            int h = (int)(t % H), n = (int)(t / H);
            int w = side ? W - 2 : 1, r = tap / 3, s = tap - 3 * r;
            int jh[3], jw[3];
            int nh = reflect_preimages(h, H, 1, jh), nw = reflect_preimages(w, W, 1, jw);
            for (int ia = 0; ia < nh; ++ia)
                for (int ib = 0; ib < nw; ++ib) {
                    int ho = jh[ia] - r, wo = jw[ib] - s;
                    if ((unsigned)ho < (unsigned)H && (unsigned)wo < (unsigned)W) {
                        float v[VEC];
                        ET<T>::unpack(ld16(dy + ((((size_t)n * H + ho) * W + wo) * K + (size_t)ch * VEC) * sizeof(T)), v);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[e] += v[e];
                    }
                }
        } else {
            int64_t t = i - npatch;
            int ch = (int)(t % cpv); t /= cpv;
            int w = (int)(t % W); t /= W;
            int which = (int)(t & 1), n = (int)(t >> 1);
            int ha = which ? H - 3 : 2, hb = which ? H - 1 : 0;
            float va[VEC], vb[VEC];
            ET<T>::unpack(ld16(dy + ((((size_t)n * H + ha) * W + w) * K + (size_t)ch * VEC) * sizeof(T)), va);
            ET<T>::unpack(ld16(dy + ((((size_t)n * H + hb) * W + w) * K + (size_t)ch * VEC) * sizeof(T)), vb);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = va[e] + vb[e];
        }
        st16(fold + (size_t)i * 16, ET<T>::pack(acc));
    }
}

// -------------------------------------------------------------------------------------------------
// Narrow-output convolution (Cout <= 16, stride 1, same-size output): the generator head, 7x7 64->3 at full
// resolution (module.py:262-264).  As an implicit GEMM it would re-read every input pixel once per tap (49x) from
// L2 for a 16-wide output tile; here the block keeps the input HALO of its 16x32-pixel output tile in LDS (one
// DMA pass, REFLECT/zero padding resolved in the DMA's per-lane source address) and all taps read shifted rows of
// it.  Weights stream through a small double-buffered LDS ring, one kernel row of taps at a time.
// -------------------------------------------------------------------------------------------------
#define HALO_TH 16
#define HALO_TW 32
#define HALO_MAXR 7

template <typename T>
__global__ __launch_bounds__(512) void conv_halo_fwd_kernel(ConvArgs a, int flip) {   // flip: weight taps mirrored (data gradient)
    constexpr int VEC = ET<T>::VEC;
    constexpr int ES = (int)sizeof(T);
    constexpr int CCH = 128 / ES;                       // channels per halo pass (one 128-byte LDS row per pixel)
    constexpr int WBUF = HALO_MAXR * 16 * 128;          // one kernel row of taps: [s][16 couts][128 B]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;                                    // [2][WBUF]
    char* sH = smem + 2 * WBUF;                         // halo [(TH+R-1)*(TW+S-1)][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index as a scalar
    const int HWd = HALO_TW + a.S - 1, HHd = HALO_TH + a.R - 1, HP = HHd * HWd;
    const int tilesW = a.W / HALO_TW, tilesH = a.H / HALO_TH;
    int b = blockIdx.x;
    const int tw = b % tilesW; b /= tilesW;
    const int th = b % tilesH;
    const int n = b / tilesH;
    const int y0 = th * HALO_TH, x0 = tw * HALO_TW;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    const int wrow = a.R * a.S * a.C;                   // weight row length (elements)
    const int pos = tid & 7, lsw = (tid >> 4) & 7;      // LDS position / swizzle key of rows (tid>>3) + 64*i
    const int lcc = pos ^ lsw;                          // logical 16-byte chunk this thread fetches

    auto stage_halo = [&](int cc) {
        for (int base = 0; base < HP; base += 64) {     // 64 halo pixels (8 per wave-instruction) per pass
            int hp = base + (tid >> 3);
            const char* src = zero;
            if (hp < HP) {
                int hy = hp / HWd, hx = hp - hy * HWd;
                int yi = y0 - a.pad_t + hy, xi = x0 - a.pad_l + hx;
                bool ok = true;
                if (a.reflect) {
                    yi = yi < 0 ? -yi : (yi >= a.H ? 2 * (a.H - 1) - yi : yi);
                    xi = xi < 0 ? -xi : (xi >= a.W ? 2 * (a.W - 1) - xi : xi);
                } else ok = (unsigned)yi < (unsigned)a.H && (unsigned)xi < (unsigned)a.W;
                if (ok) src = a.src + ((((size_t)n * a.H + yi) * a.W + xi) * a.C + cc * CCH) * ES + lcc * 16;
            }
            if (base + wave * 8 < HP + 8)                // wave-uniform: skip instructions wholly past the halo
                dma16_to_lds(src, (__attribute__((address_space(3))) void*)(sH + (base + wave * 8) * 128));
        }
    };
    auto stage_weights = [&](int buf, int r, int cc) {   // taps (r, 0..S-1): rows = s*16 + cout
        for (int base = 0; base < a.S * 16; base += 64) {
            int row = base + (tid >> 3);
            const char* src = zero;
            if (row < a.S * 16) {
                int s = row >> 4, k = row & 15;
                const int tap = flip ? (a.R - 1 - r) * a.S + (a.S - 1 - s) : r * a.S + s;
                if (k < a.K) src = a.wmat + ((size_t)k * wrow + (size_t)tap * a.C + cc * CCH) * ES + lcc * 16;
            }
            if (base + wave * 8 < a.S * 16)
                dma16_to_lds(src, (__attribute__((address_space(3))) void*)(sW + buf * WBUF + (base + wave * 8) * 128));
        }
    };

    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;

    const int nchunks = a.C / CCH;
    for (int cc = 0; cc < nchunks; ++cc) {
        __syncthreads();                                 // previous pass has finished reading the halo
        stage_halo(cc);
        stage_weights(0, 0, cc);
        SGG_WAIT_VM0();
        __syncthreads();
        for (int r = 0; r < a.R; ++r) {
            const int cur = r & 1;
            if (r + 1 < a.R) stage_weights(cur ^ 1, r + 1, cc);
            for (int s = 0; s < a.S; ++s) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int wr = s * 16 + frow;
                    u32x4 fw = ld16(sW + cur * WBUF + wr * 128 + ((((fq + 4 * kk) ^ ((wr >> 1) & 7))) << 4));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int y = 2 * wave + (j >> 1), x = (j & 1) * 16 + frow;
                        const int hp = (y + r) * HWd + x + s;
                        u32x4 fp = ld16(sH + hp * 128 + ((((fq + 4 * kk) ^ ((hp >> 1) & 7))) << 4));
                        if constexpr (sizeof(T) == 2) {
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw), __builtin_bit_cast(bf16x8, fp), acc[j], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw[e]), __uint_as_float(fp[e]), acc[j], 0, 0, 0);
                        }
                    }
                }
            }
            SGG_WAIT_VM0();
            __syncthreads();
        }
    }

    // D[cout = 4*fq + e][pixel = frow]
    const int dc = fq * 4;
    if (dc < a.K) {
        float bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = a.bias ? a.bias[dc + e] : 0.f;
        act_dispatch(a.act, [&](auto act_c) {
            constexpr int ACT = decltype(act_c)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = y0 + 2 * wave + (j >> 1), x = x0 + (j & 1) * 16 + frow;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_apply_c<ACT>(acc[j][e] + bv[e], a.leak);
                T* o = reinterpret_cast<T*>(a.dst) + (((size_t)n * a.H + y) * a.W + x) * a.K + dc;
                if constexpr (sizeof(T) == 2) {
                    bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(o) = pk;
                } else *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
            }
        });
    }
}

// -------------------------------------------------------------------------------------------------
// 7x7, 64 -> 3 channels at full resolution (bf16): the generator head forward (module.py:262-264) and, with mirrored
// taps, the main part of the stem's data gradient (module.py:230-232 backward).
//
// conv_halo_fwd_kernel above puts the 3 output channels in a 16-wide MFMA dimension (13 of 16 columns wasted) and
// reads one LDS fragment per MFMA.  Here the GEMM N dimension carries (output channel, tap COLUMN) = 3 x 7 = 21 -> 32
// and only the tap ROW stays in the reduction:
//     D[q][(co, s)] = sum_{r, c} halo[y + r][q][c] * w[r][s][c][co]          (q = halo column, K = 7 x 64)
//     out[y][p][co] = sum_s D[p + s][(co, s)]                                (a shifted sum, done through LDS)
// i.e. 140 MFMAs per 64 output pixels instead of 392, and one fragment read per 2 MFMAs.  The weight fragments (28 per
// lane) live in registers for the whole block.  One block = 8 output rows x 64 columns, one wave per row; the 14 x 70
// pixel input halo (128 B per pixel) is DMA'd once, REFLECT / zero padding resolved in the per-lane source address.
// -------------------------------------------------------------------------------------------------
#define N7_TH 8
#define N7_TW 64
#define N7_PITCH 72                                     // halo row pitch in pixels (70 used; 9 DMAs of 8)
#define N7_ROWS (N7_TH + 6)
#define N7_LDS (N7_ROWS * N7_PITCH * 128 + 2048)
#define N7_ZP 68                                        // floats per (co, s) row of the shifted-sum buffer

struct N7Args {
    const char* src;     // (N,H,W,64) bf16
    const char* wmat;    // [dest channel][49 taps][64] bf16 (w_fwd of the head / w_dgrad of the stem)
    const float* bias;   // per dest channel or nullptr
    void* dst;           // bf16 (N,Ho,Wo,8)  or  f32 (N,Ho,Wo,4) when dst_f32
    int N, H, W, Ho, Wo, pt, pl, K, reflect, flip, act, dst_f32;
    float leak;
};

__global__ __launch_bounds__(512) void conv7_narrow_out_kernel(N7Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lH = (lds_char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesW = (a.Wo + N7_TW - 1) / N7_TW, tilesH = (a.Ho + N7_TH - 1) / N7_TH;
    const int ntiles = a.N * tilesH * tilesW;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    const int frow = lane & 15, fq = lane >> 4;
    // ---- weight fragments into registers, once per (persistent) block: 28 x 16 B per lane.  B operand: lane supplies
    // k = 8*(lane/16).., column n = nt*16 + lane%16
    u32x4 wr[7][2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int nn = nt * 16 + frow, co = nn >> 3, sc = nn & 7;
        const bool ok = co < a.K && co < 3 && sc < 7;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int tap = a.flip ? (6 - r) * 7 + (6 - sc) : r * 7 + sc;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                wr[r][kk][nt] = ok ? ld16(a.wmat + (((size_t)co * 49 + tap) * 64 + kk * 32 + fq * 8) * 2) : zero16();
        }
    }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int b = tile;
    const int tw = b % tilesW; b /= tilesW;
    const int th = b % tilesH;
    const int n = b / tilesH;
    const int y0 = th * N7_TH, x0 = tw * N7_TW;
    __syncthreads();                                    // the previous tile's shifted-sum buffer (in the halo region) is done

    // ---- halo DMA: 14 rows x 9 instructions of 8 pixels; instruction id = row * 9 + i, dealt round-robin to the waves
    {
        const int hsub = lane >> 3, hpos = lane & 7;
        for (int id = wave; id < N7_ROWS * 9; id += 8) {
            const int hy = id / 9, hx = (id - hy * 9) * 8 + hsub;
            int yi = y0 - a.pt + hy, xi = x0 - a.pl + hx;
            bool ok = hx < N7_TW + 6;
            if (a.reflect) {
                yi = yi < 0 ? -yi : (yi >= a.H ? 2 * (a.H - 1) - yi : yi);
                xi = xi < 0 ? -xi : (xi >= a.W ? 2 * (a.W - 1) - xi : xi);
                ok = ok && (unsigned)yi < (unsigned)a.H && (unsigned)xi < (unsigned)a.W;   // (tiles past the image edge)
            } else ok = ok && (unsigned)yi < (unsigned)a.H && (unsigned)xi < (unsigned)a.W;
            const int hp = hy * N7_PITCH + hx;
            const int key = (hp >> 1) & 7;
            const char* src = ok ? a.src + ((((size_t)n * a.H + yi) * a.W + xi) * 64) * 2 + ((hpos ^ key) << 4) : zero;
            dma16_to_lds(src, (__attribute__((address_space(3))) void*)(lH + (hy * N7_PITCH + (id - hy * 9) * 8) * 128));
        }
    }
    f32x4 acc[5][2];
#pragma unroll
    for (int qf = 0; qf < 5; ++qf)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[qf][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    SGG_WAIT_VM0();
    __syncthreads();

    // ---- D[q][(co,s)] for this wave's output row: A operand = halo pixels (lane: pixel frow, chunk fq), B = weights
    const char* sH = smem;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const int hrow = wave + r;
        const int key = ((hrow * 4) + (frow >> 1)) & 7;             // (hp >> 1) & 7 with hp = hrow*72 + qf*16 + frow
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const char* base = sH + ((size_t)(hrow * N7_PITCH + frow)) * 128 + (((kk * 4 + fq) ^ key) << 4);
#pragma unroll
            for (int qf = 0; qf < 5; ++qf) {
                const u32x4 px = ld16(base + qf * 16 * 128);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[qf][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, px), __builtin_bit_cast(bf16x8, wr[r][kk][nt]),
                                                                          acc[qf][nt], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                                 // every wave is done with the halo: reuse it for Z

    // ---- shifted sum: Z[(co,s)][p] = D[p + s][(co,s)], then out[p][co] = sum_s Z[(co,s)][p] (fixed order)
    float* Z = reinterpret_cast<float*>(smem) + (size_t)wave * (21 * N7_ZP);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int nn = nt * 16 + frow, co = nn >> 3, sc = nn & 7;
        if (co < 3 && sc < 7) {
#pragma unroll
            for (int qf = 0; qf < 5; ++qf)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int pcol = qf * 16 + fq * 4 + e - sc;
                    if ((unsigned)pcol < (unsigned)N7_TW) Z[(co * 7 + sc) * N7_ZP + pcol] = acc[qf][nt][e];
                }
        }
    }
    SGG_WAIT_LGKM0();
    __builtin_amdgcn_wave_barrier();
    const int y = y0 + wave, x = x0 + lane;
    float o[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) {
        float v = 0.f;
#pragma unroll
        for (int sc = 0; sc < 7; ++sc) v += Z[(co * 7 + sc) * N7_ZP + lane];
        o[co] = (co < a.K) ? v + (a.bias ? a.bias[co] : 0.f) : 0.f;
    }
    act_dispatch(a.act, [&](auto act_c) {
#pragma unroll
        for (int co = 0; co < 3; ++co) if (co < a.K) o[co] = act_apply_c<decltype(act_c)::value>(o[co], a.leak);
    });
    if (y < a.Ho && x < a.Wo) {
        const size_t pix = ((size_t)n * a.Ho + y) * a.Wo + x;
        if (a.dst_f32) {
            reinterpret_cast<f32x4*>(a.dst)[pix] = (f32x4){o[0], o[1], o[2], 0.f};
        } else {
            u32x4 pk = zero16();
            const bf16 b0 = (bf16)o[0], b1 = (bf16)o[1], b2 = (bf16)o[2];
            pk[0] = (uint32_t)__builtin_bit_cast(uint16_t, b0) | ((uint32_t)__builtin_bit_cast(uint16_t, b1) << 16);
            pk[1] = (uint32_t)__builtin_bit_cast(uint16_t, b2);
            st16(reinterpret_cast<char*>(a.dst) + pix * 16, pk);
        }
    }
  }
}

// MirrorPadGrad for REFLECT pad 3 (tf.pad backward in front of the stem, module.py:230): dxp is the data gradient on the
// PADDED grid (N, H+6, W+6, 4) f32; every image pixel sums its (up to 2 x 2) mirror pre-images in fixed order and is
// rounded to bf16 once.
// addend (nullable, (N,H,W,8) bf16 like dx): added before the one rounding (a gradient join folded into the store).
__global__ __launch_bounds__(256) void pad3_fold_kernel(const f32x4* dxp, const char* addend, char* dx, int N, int H, int W) {
    const int64_t total = (int64_t)N * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const int64_t t = i / W;
        const int h = (int)(t % H), n = (int)(t / H);
        int jh[3], jw[3];
        const int nh = reflect_preimages(h, H, 3, jh), nw = reflect_preimages(w, W, 3, jw);
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        for (int ih = 0; ih < nh; ++ih)
            for (int iw = 0; iw < nw; ++iw) {
                const f32x4 d = dxp[((size_t)n * (H + 6) + jh[ih]) * (W + 6) + jw[iw]];
                v0 += d[0]; v1 += d[1]; v2 += d[2];
            }
        if (addend) {
            const bf16x4 ad = *reinterpret_cast<const bf16x4*>(addend + (size_t)i * 16);
            v0 += (float)ad[0]; v1 += (float)ad[1]; v2 += (float)ad[2];
        }
        u32x4 pk = zero16();
        const bf16 b0 = (bf16)v0, b1 = (bf16)v1, b2 = (bf16)v2;
        pk[0] = (uint32_t)__builtin_bit_cast(uint16_t, b0) | ((uint32_t)__builtin_bit_cast(uint16_t, b1) << 16);
        pk[1] = (uint32_t)__builtin_bit_cast(uint16_t, b2);
        st16(dx + (size_t)i * 16, pk);
    }
}

// -------------------------------------------------------------------------------------------------
// Weight gradient of the two 7x7 layers with a 3-channel side (bf16): the head (64 -> 3, module.py:262-264) and the
// stem (3 -> 64, module.py:230-232).  Both are
//     G[r][s][c][j] = sum_{n, py, px} A~[n][py + r - pa][px + s - pa][c] * B[n][py][px][j],     c < 64, j < 3
// with A the 64-channel tensor (head: the layer input, REFLECT padded; stem: dy, zero padded by 6 -- the sum then runs
// over the PADDED grid, B = the explicitly reflect-padded input, and taps come out mirrored) and B the 3-channel one.
// conv_halo_wgrad_kernel gives every tap its own MFMAs with 13 of 16 columns empty (196 per 32 pixels, two LDS reads
// each).  Here the GEMM N dimension carries (j, tap column s) = 3 x 7 -> 32 and each wave owns ONE tap row r:
//     D_r[c][(j,s)] += sum_q A~[py + r][q][c] * Bs[py][q][(j,s)],      Bs[py][q][(j,s)] = B[py][q - s][j]
// (q = padded column inside a 58-pixel strip: 64 values = two k-steps), 8 MFMAs per 32 pixels and wave, 12 LDS reads.
// Bs -- an im2col of the 3-channel tensor along the row, 64 B per pixel -- is built in LDS by the eighth wave while
// waves 0..6 multiply.  A block walks down a strip 4 rows at a time: the 4 new A rows are DMA'd into a 14-row ring
// while the other 10 are in use.  Both operands are read with the transposing ds_read_b64_tr_b16 (pixels are the
// reduction index).  One f32 slab [7][64][32] per block; wgrad7_reduce_kernel sums them in fixed order.
// -------------------------------------------------------------------------------------------------
#ifndef W7_INTERLEAVE
#define W7_INTERLEAVE 1                                // the main loop's A-row DMAs between the MFMA batches (0: all at the head of the step)
#endif
#define W7_TW 58                                        // strip width in pixels of B's grid (58 + 6 = 64 = two k-steps)
#define W7_AROW (64 * 128)                              // bytes per A row (64 pixels x 64 channels)
#define W7_BSROW (64 * 64)                              // bytes per Bs row (64 pixels x 32 columns)
// RS rows per step; A rows are DMA'd PD steps ahead (they come from HBM: ~2 us under load, a step is < 1 us) into a
// ring of RS + 6 + PD*RS rows; raw B rows three steps ahead into a ring of 4 steps; Bs double-buffered.
template <int RS, int PD> struct W7Cfg {
    static constexpr int RING = RS + 6 + PD * RS;
    static constexpr int OFF_BS = RING * W7_AROW;
    static constexpr int OFF_BRAW = OFF_BS + 2 * RS * W7_BSROW;
    static constexpr int LDS = OFF_BRAW + 4 * RS * 512 + 1024;
    static constexpr int B_PER_STEP = RS * 2;           // raw-B DMA instructions per step (wave 7)
    static constexpr int NISS = 6;                      // waves 0..5 issue the main loop's A-row DMAs (wave 7 has the B side)
    static constexpr int A_PER_WAVE = RS * 8 / NISS;    // RS rows x 8 instructions over the issuer waves
    static_assert(RS * 8 % NISS == 0, "issuer waves must carry equal DMA counts (the vmcnt bookkeeping relies on it)");
};

struct W7Args {
    const char* A;       // (N,HA,WA,64) bf16
    const char* B;       // (N,HB,WB,8) bf16 -- the summation grid
    float* slabs;        // [blocks][7][64][32] f32
    int N, HA, WA, HB, WB, pa, reflectA, nstrips, nseg, rows_per_seg;
    int ablate;          // lab build only (SGG_ABLATE): 1 no fragment reads / MFMAs, 2 no Bs build, 3 no A-row DMA in the loop
};

template <int N> __device__ inline void w7_wait_vm() { sgg_wait_vm<N>(); }

template <int RS, int PD>
__global__ __launch_bounds__(512) void wgrad7_kernel(W7Args a) {
    using Cfg = W7Cfg<RS, PD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds = (lds_char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    int b = blockIdx.x;
    const int seg = b % a.nseg; b /= a.nseg;
    const int strip = b % a.nstrips;
    const int n = b / a.nstrips;
    const int yb = seg * a.rows_per_seg;
    const int ye = yb + a.rows_per_seg < a.HB ? yb + a.rows_per_seg : a.HB;
    const int x0 = strip * W7_TW;
    const int nsteps = (ye - yb + RS - 1) / RS;

    // ---- A rows: ring row k (k = 0 .. rows + 5) is image row yb + k - pa; 8 DMA instructions of 8 pixels per row.
    // Rows k0 .. k0+nrows-1; instruction ids are dealt round-robin to the issuing waves.  Rows past the image read the zero page.
    const int hsub = lane >> 3, hpos = lane & 7;
    auto load_a_rows = [&](int k0, int nrows, int vw, int nw, int j0 = 0, int j1 = 1 << 20) {   // this wave acts as issuer vw of nw; its pieces [j0, j1)
        for (int id = vw + nw * j0, j = j0; id < nrows * 8 && j < j1; id += nw, ++j) {
            const int k = k0 + (id >> 3), i = id & 7;
            const int q = i * 8 + hsub;
            int yi = yb + k - a.pa, xi = x0 - a.pa + q;
            if (a.reflectA) {
                yi = yi < 0 ? -yi : (yi >= a.HA ? 2 * (a.HA - 1) - yi : yi);
                xi = xi < 0 ? -xi : (xi >= a.WA ? 2 * (a.WA - 1) - xi : xi);
            }
            const bool ok = (unsigned)yi < (unsigned)a.HA && (unsigned)xi < (unsigned)a.WA;
            const int key = ((q >> 1) & 1) | (((q >> 3) & 1) << 1);           // keyed on pixel bits 1 and 3: the transposed reads are conflict-free
            const char* src = ok ? a.A + ((((size_t)n * a.HA + yi) * a.WA + xi) * 64) * 2 + ((hpos ^ (key << 1)) << 4) : zero;
            dma16_to_lds(src, (__attribute__((address_space(3))) void*)(lds + (k % Cfg::RING) * W7_AROW + i * 1024));
        }
    };
    // ---- B rows of a step (wave 7): raw row DMA, then the im2col Bs[q][(j,s)] = B[q - s][j]
    auto load_b_raw = [&](int step) {                    // one lane = one pixel; a 4-byte wave-instruction moves 64 x 4 B:
        for (int i = 0; i < RS; ++i) {                   // dword 0 = channels (0,1), dword 1 = (2,3)
            const int py = yb + step * RS + i, px = x0 + lane;
            const bool ok = py < ye && lane < W7_TW && px < a.WB;
            const char* src = ok ? a.B + (((size_t)n * a.HB + py) * a.WB + px) * 16 : zero;
#pragma unroll
            for (int part = 0; part < 2; ++part)
                dma4_to_lds(src + part * 4, (__attribute__((address_space(3))) void*)(lds + Cfg::OFF_BRAW + ((step & 3) * RS + i) * 512 + part * 256));
        }
    };
    auto build_bs = [&](int step) {
        char* bs = smem + Cfg::OFF_BS + (step & 1) * (RS * W7_BSROW);
        const int q = lane;
        const int bsw = ((q >> 3) & 1) << 1;                                   // chunk swizzle of the Bs rows
        for (int i = 0; i < RS; ++i) {
            const char* raw = smem + Cfg::OFF_BRAW + ((step & 3) * RS + i) * 512;   // [part 0..1][pixel] dwords
            uint32_t lo[7], hi[7];                                             // pixel q - s: channels (0,1) and (2,3)
            // (unconditional loads from a clamped index + a select: a conditional load becomes a branch with a full
            // lgkmcnt(0) wait behind every read, ~2 000 cycles per row)
#pragma unroll
            for (int sc = 0; sc < 7; ++sc) {
                const int idx = q - sc, ic = idx < 0 ? 0 : idx;
                lo[sc] = *reinterpret_cast<const uint32_t*>(raw + 4 * ic);
                hi[sc] = *reinterpret_cast<const uint32_t*>(raw + 256 + 4 * ic);
            }
#pragma unroll
            for (int sc = 0; sc < 7; ++sc) {
                const bool ok = q - sc >= 0;
                lo[sc] = ok ? lo[sc] : 0u;
                hi[sc] = ok ? hi[sc] : 0u;
            }
            // columns n = j*8 + s: 16-byte chunk j holds s = 0..6 (+ a zero) for channel j; chunk 3 is zero
            u32x4 ch[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                uint32_t e[8];
#pragma unroll
                for (int sc = 0; sc < 7; ++sc) {
                    const uint32_t w = j < 2 ? lo[sc] : hi[sc];
                    e[sc] = (j == 1) ? (w >> 16) : (w & 0xffffu);
                }
                e[7] = 0u;
                ch[j] = (u32x4){e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
            }
            char* row = bs + i * W7_BSROW + q * 64;
            st16(row + ((0 ^ bsw) << 4), ch[0]);
            st16(row + ((1 ^ bsw) << 4), ch[1]);
            st16(row + ((2 ^ bsw) << 4), ch[2]);
            st16(row + ((3 ^ bsw) << 4), zero16());
        }
    };

    f32x4 acc[4][2];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) acc[it][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue (all 8 waves issue, everything is drained): A rows of steps 0 .. PD-1, raw B of steps 0 and 1, Bs of step 0
    if (wave == 7) load_b_raw(0);
    load_a_rows(0, RS + 6 + (PD - 1) * RS, wave, 8);
    SGG_WAIT_VM0();
    if (wave == 7) {
        build_bs(0);
        load_b_raw(1);
        load_b_raw(2);
    }
    SGG_WAIT_LGKM0();
    __syncthreads();

    // Transposed fragment reads: lane 4qq+pp of 16-lane group g supplies row (pixel) 8g+qq [+4 for the upper half], columns
    // 4pp..4pp+3 of the 16-column tile.  Everything lane-dependent is folded into six LDS byte addresses computed once
    // (the swizzle keys depend only on pixel bits 1 and 3 = qq bit 1 and g bit 0); a read is then base + scalar + immediate.
    const int g = lane >> 4, u = lane & 15, qq = u >> 2, pp = u & 3;
    const uint32_t lb = (uint32_t)(uintptr_t)lds;
    const int keyA = ((qq >> 1) & 1) | ((g & 1) << 1);
    uint32_t offA[4], offB[2];
#pragma unroll
    for (int it = 0; it < 4; ++it) offA[it] = lb + (8 * g + qq) * 128 + ((((it ^ keyA) << 1) | (pp >> 1)) << 4) + (pp & 1) * 8;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) offB[jt] = lb + Cfg::OFF_BS + (8 * g + qq) * 64 + ((((jt * 2) | (pp >> 1)) ^ ((g & 1) << 1)) << 4) + (pp & 1) * 8;
    auto tr = [](uint32_t addr) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(uintptr_t)addr); };
    for (int t = 0; t < nsteps; ++t) {
        // rows of step t + PD into the slots step t - 1 released, by waves 0..5 (always issued: rows past the segment are
        // harmless and keep every issuer's count of outstanding DMAs per step constant)
        if (!W7_INTERLEAVE && wave < Cfg::NISS && SGG_ABLATE_OF(a) != 3) load_a_rows(RS + 6 + (t + PD - 1) * RS, RS, wave, Cfg::NISS);
        if (wave < 7 && SGG_ABLATE_OF(a) == 1) {
        } else if (wave < 7) {
            const int r = wave;
            // RS x 2 batches of (2 B + 4 A fragments -> 8 MFMAs); the fragments of batch b+1 are requested before the MFMAs
            // of batch b are issued (a batch's reads would otherwise sit exposed: ~200 cycles of LDS latency per 128 of MFMA)
            bf16x4 fbq[2][2][2], faq[2][4][2];            // [buffer][tile][lo/hi]
            auto request = [&](int bi, int buf) {
                const int i = bi >> 1, ks = bi & 1;
                const uint32_t abase = (uint32_t)((t * RS + i + r) % Cfg::RING) * W7_AROW + ks * 32 * 128;
                const uint32_t bbase = (uint32_t)((t & 1) * RS + i) * W7_BSROW + ks * 32 * 64;
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) { fbq[buf][jt][0] = tr(offB[jt] + bbase); fbq[buf][jt][1] = tr(offB[jt] + bbase + 4 * 64); }
#pragma unroll
                for (int it = 0; it < 4; ++it) { faq[buf][it][0] = tr(offA[it] + abase); faq[buf][it][1] = tr(offA[it] + abase + 4 * 128); }
            };
            request(0, 0);
#pragma unroll
            for (int bi = 0; bi < RS * 2; ++bi) {
                const int buf = bi & 1;
                if (bi + 1 < RS * 2) request(bi + 1, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 fb[2];
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    const bf16x4 lo = fbq[buf][jt][0], hi = fbq[buf][jt][1];
                    fb[jt] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const bf16x4 lo = faq[buf][it][0], hi = faq[buf][it][1];
                    const bf16x8 fa = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt) acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[jt], acc[it][jt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // the A-row DMAs of step t + PD, ONE piece per issuer wave after each of its first MFMA batches instead of all of
                // them at the head of the step: a piece costs its wave 60-180 cycles of issue, which now fall under the SIMD
                // partner's MFMAs (with all six issuers at the head of the step the matrix pipes idled for that long every step)
                if (W7_INTERLEAVE && bi < Cfg::A_PER_WAVE && wave < Cfg::NISS && SGG_ABLATE_OF(a) != 3) {
                    load_a_rows(RS + 6 + (t + PD - 1) * RS, RS, wave, Cfg::NISS, bi, bi + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the rows of step t+1 (issued PD-1 steps ago) have landed; the later steps' may still be in flight
            if (wave < Cfg::NISS) w7_wait_vm<Cfg::A_PER_WAVE * (PD - 1)>();
        } else {
            // wave 7: raw B of step t+1 (issued TWO steps ago; only its own B DMAs are in its queue, step t+2's may still
            // fly) -> Bs(t+1) while the others multiply; then the raw B of step t+3 goes out.  (With the raw rows requested
            // only one step ahead this wave sat out a full HBM latency per step, and the block with it.)
            w7_wait_vm<Cfg::B_PER_STEP>();
            if (SGG_ABLATE_OF(a) != 2) build_bs(t + 1);
            load_b_raw(t + 3);
        }
        SGG_WAIT_LGKM0();
        __syncthreads();
    }
    SGG_WAIT_VM0();     // nothing may still be writing LDS when the block retires

    // D_r[row = it*16 + 4g + e -> channel c][col = jt*16 + u -> (j,s)]
    if (wave < 7) {
        float* slab = a.slabs + ((size_t)blockIdx.x * 7 + wave) * 64 * 32;
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int e = 0; e < 4; ++e) slab[(it * 16 + 4 * g + e) * 32 + jt * 16 + u] = acc[it][jt][e];
    }
}

// dw[...] (+)= sum over blocks of slab[b][r][c][j*8 + s], in fixed order.  mode 0 (head): dw is HWIO (7,7,64,Kr):
// dw[r][s][c][j].  mode 1 (stem): dw is (7,7,Cr,64) and the taps come out mirrored: dw[6-r][6-s][j][c].
__global__ __launch_bounds__(1024) void wgrad7_reduce_kernel(const float* slabs, int nblocks, float* dw, int mode, int nj, int accumulate) {
    __shared__ float red[32][32];
    const int o = threadIdx.x & 31, part = threadIdx.x >> 5;     // 32 outputs x 32 slab lanes
    const int idx = blockIdx.x * 32 + o;                 // (r, c, col) with col = j*8 + s in 0..31
    const int col = idx & 31, c = (idx >> 5) & 63, r = idx >> 11;
    float v = 0.f;
    for (int b0 = part; b0 < nblocks; b0 += 32 * 4) {    // four independent loads in flight, added in slab order
        float x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = b0 + 32 * k;
            x[k] = b < nblocks ? slabs[(size_t)b * (7 * 64 * 32) + idx] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v += x[k];
    }
    red[part][o] = v;
    __syncthreads();
    if (part == 0) {
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < 32; ++p) t += red[p][o];
        const int j = col >> 3, sc = col & 7;
        if (j < nj && sc < 7) {
            float* d = mode == 0 ? dw + ((size_t)(r * 7 + sc) * 64 + c) * nj + j
                                 : dw + ((size_t)((6 - r) * 7 + (6 - sc)) * nj + j) * 64 + c;
            *d = accumulate ? *d + t : t;
        }
    }
}

// explicit REFLECT pad 3 of an 8-channel bf16 tensor (the stem's input, as wgrad7_kernel's B operand): (N,H,W,8) -> (N,H+6,W+6,8)
__global__ __launch_bounds__(256) void reflect_pad3_kernel(const char* x, char* xp, int N, int H, int W) {
    const int Hp = H + 6, Wp = W + 6;
    const int64_t total = (int64_t)N * Hp * Wp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int wp = (int)(i % Wp);
        const int64_t t = i / Wp;
        const int hp = (int)(t % Hp), n = (int)(t / Hp);
        int h = hp - 3, w = wp - 3;
        h = h < 0 ? -h : (h >= H ? 2 * (H - 1) - h : h);
        w = w < 0 ? -w : (w >= W ? 2 * (W - 1) - w : w);
        st16(xp + (size_t)i * 16, ld16(x + (((size_t)n * H + h) * W + w) * 16));
    }
}

static bool w7_head_ok(const sgg_conv_desc* d) {        // 64 -> <= 3 (stored 8), 7x7, pad 3
    return sgg_config().n7 && d->dtype == SGG_BF16 && d->R == 7 && d->S == 7 && d->stride == 1 && d->C == 64 && d->K == 8 &&
           d->Ho == d->H && d->Wo == d->W && d->pad_t == 3 && d->pad_l == 3 && d->H >= 8 && d->W >= 8;
}
static bool w7_stem_ok(const sgg_conv_desc* d) {        // <= 3 (stored 8) -> 64, 7x7, REFLECT 3
    return sgg_config().n7 && d->dtype == SGG_BF16 && d->R == 7 && d->S == 7 && d->stride == 1 && d->C == 8 && d->K == 64 &&
           d->Ho == d->H && d->Wo == d->W && d->pad_t == 3 && d->pad_l == 3 && d->pad_mode == SGG_PAD_REFLECT && d->H >= 8 && d->W >= 8;
}
#ifndef W7_RSTEP
#define W7_RSTEP 3
#define W7_PDIST 2
#endif
struct W7Plan { int HB, WB, nstrips, nseg, rows_per_seg, blocks; size_t slab_bytes, xpad_bytes; };
static W7Plan w7_plan(const sgg_conv_desc* d, bool stem) {
    W7Plan p;
    p.HB = d->H + (stem ? 6 : 0); p.WB = d->W + (stem ? 6 : 0);
    p.nstrips = (p.WB + W7_TW - 1) / W7_TW;
    int nseg = 512 / (d->N * p.nstrips);                 // ~2 blocks per CU (one resident at a time: 150 KB of LDS)
    if (nseg < 1) nseg = 1;
    int rps = (p.HB + nseg - 1) / nseg;
    rps = (rps + W7_RSTEP - 1) / W7_RSTEP * W7_RSTEP;
    if (rps < 8) rps = 8;
    p.rows_per_seg = rps;
    p.nseg = (p.HB + rps - 1) / rps;
    p.blocks = d->N * p.nstrips * p.nseg;
    p.slab_bytes = (size_t)p.blocks * 7 * 64 * 32 * sizeof(float);
    p.xpad_bytes = stem ? align_up((size_t)d->N * p.HB * p.WB * 16, 256) : 0;
    return p;
}
static int run_w7(const sgg_conv_desc* d, bool stem, const void* x, const void* dy, float* dw, int Cr, int Kr, int accumulate,
                  void* ws, size_t ws_bytes, hipStream_t s) {
    const W7Plan p = w7_plan(d, stem);
    if (!ws || ws_bytes < p.slab_bytes + p.xpad_bytes) return SGG_EWORKSPACE;
    if ((stem ? Cr : Kr) > 3) return SGG_EUNSUPPORTED;
    W7Args a;
    a.slabs = (float*)ws;
    a.N = d->N; a.HB = p.HB; a.WB = p.WB; a.nstrips = p.nstrips; a.nseg = p.nseg; a.rows_per_seg = p.rows_per_seg;
    a.HA = d->H; a.WA = d->W; a.ablate = sgg_config().ablate;
    if (stem) {
        char* xpad = (char*)ws + p.slab_bytes;
        const int64_t total = (int64_t)d->N * p.HB * p.WB;
        int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(reflect_pad3_kernel, dim3(blocks), dim3(256), 0, s, (const char*)x, xpad, d->N, d->H, d->W);
        a.A = (const char*)dy; a.B = xpad; a.pa = 6; a.reflectA = 0;
    } else {
        a.A = (const char*)x; a.B = (const char*)dy; a.pa = 3; a.reflectA = d->pad_mode == SGG_PAD_REFLECT;
    }
    auto kern = wgrad7_kernel<W7_RSTEP, W7_PDIST>;
    constexpr int lds = W7Cfg<W7_RSTEP, W7_PDIST>::LDS;
    static_assert(lds <= 160 * 1024, "wgrad7 LDS budget");
    SGG_LDS_ATTR(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)p.blocks), dim3(512), lds, s, a);
    int rc = sgg_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad7_reduce_kernel, dim3(7 * 64 * 32 / 32), dim3(1024), 0, s, (const float*)ws, p.blocks, dw, stem ? 1 : 0, stem ? Cr : Kr, accumulate);
    return sgg_check_launch();
}

// head forward: 7x7 stride 1, 64 -> <= 3 channels (stored in 8), bf16, same-size output
// -------------------------------------------------------------------------------------------------
// Data gradient of the discriminator's first conv (module.py:284: 3x3, stride 2, SAME, 3(8) -> 64 channels): dx has 8 channels,
// dy 64.  As an implicit GEMM (conv_gemm_glds_kernel<DGRAD, 256x16>) the layer ran at 11 TFLOP/s, 162 us for both discriminators'
// 16 images -- all of it gather overhead: the work is 0.45 GMAC against 50 MB of traffic (~10 us).  Here a wave owns two dx rows
// (2m, 2m+1) x 32 columns (both column parities of 16 column pairs j): with even H, W (TF 'SAME' pads bottom / right only) the
// stride-2 taps that reach them are
//     row 2m  : r = 0 from dy row m,  r = 2 from dy row m-1        column 2j  : s = 0 from dy column j,  s = 2 from column j-1
//     row 2m+1: r = 1 from dy row m                                 column 2j+1: s = 1 from dy column j
// i.e. four fragments of 16 dy pixels (rows m-1 / m, columns j-1.. / j..), loaded straight from global memory in the MFMA operand
// layout (a pixel's 64 channels are the reduction: 2 k-steps), feed 9 tap products: D[channel][pixel] += W_tap[channel][k] dy[k][pixel].
// The 9 x 2 weight fragments stay in registers while a wave walks its share of the work items.  No LDS, no barrier.
// Two networks on one stacked batch: images >= nsplit take w2.
// -------------------------------------------------------------------------------------------------
struct S2NArgs {
    const char* dy;      // (N,Ho,Wo,64) bf16
    const char* w;       // w_dgrad [8][9*64] bf16 (rows = dx channels)
    const char* w2;
    char* dx;            // (N,H,W,8) bf16
    const char* addend;  // nullable, like dx: added to the result before its one rounding (a gradient join folded into the store)
    int N, nsplit, H, W, Ho, Wo, items, items_per_wave;
};

template <bool ADD>
__global__ __launch_bounds__(256) void conv3x3s2_narrow_dgrad_kernel(S2NArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int frow = lane & 15, fq = lane >> 4;
    const int gw = blockIdx.x * 4 + wave;
    const int it0 = gw * a.items_per_wave, it1 = min(a.items, it0 + a.items_per_wave);
    const int jblocks = (a.Wo + 15) >> 4, mrows = a.Ho;
    int cur_net = -1;
    u32x4 wr[9][2];
    for (int it = it0; it < it1; ++it) {
        int b = it;
        const int jb = b % jblocks; b /= jblocks;
        const int m = b % mrows;
        const int n = b / mrows;
        const int net = n >= a.nsplit ? 1 : 0;
        if (net != cur_net) {                               // (a wave's items are consecutive: at most one switch)
            cur_net = net;
            const char* wm = net ? a.w2 : a.w;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    wr[t][kk] = frow < 8 ? ld16(wm + (((size_t)frow * 9 + t) * 64 + kk * 32 + fq * 8) * 2) : zero16();
        }
        // dy fragments: [row m-1 / m][column shift -1 / 0][k-step]
        u32x4 fy[2][2][2];
        const int j = jb * 16 + frow;
#pragma unroll
        for (int ry = 0; ry < 2; ++ry)
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                const int yo = m - 1 + ry, xo = j - 1 + sx;
                const bool ok = yo >= 0 && (unsigned)xo < (unsigned)a.Wo;
                const char* src = a.dy + ((((size_t)n * a.Ho + (ok ? yo : 0)) * a.Wo + (ok ? xo : 0)) * 64 + fq * 8) * 2;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fy[ry][sx][kk] = ok ? ld16(src + kk * 64) : zero16();
            }
        f32x4 acc[2][2];                                    // [dx row parity][dx column parity]
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) acc[py][px] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto mac = [&](int py, int px, int r, int sx_tap, int ry, int sx) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                acc[py][px] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wr[r * 3 + sx_tap][kk]),
                                                                      __builtin_bit_cast(bf16x8, fy[ry][sx][kk]), acc[py][px], 0, 0, 0);
        };
        // (dx row parity, dx column parity) <- tap (r, s) from fragment (dy row m-1+ry, column shift sx); fixed order
        mac(0, 0, 0, 0, 1, 1); mac(0, 0, 0, 2, 1, 0); mac(0, 0, 2, 0, 0, 1); mac(0, 0, 2, 2, 0, 0);
        mac(0, 1, 0, 1, 1, 1); mac(0, 1, 2, 1, 0, 1);
        mac(1, 0, 1, 0, 1, 1); mac(1, 0, 1, 2, 1, 0);
        mac(1, 1, 1, 1, 1, 1);
        // D[channel = 4 fq + e][pixel = frow]: lanes fq < 2 hold the 8 channels of column pair j
        if (fq < 2 && j < a.Wo) {
            bf16x4 ad[2][2];
            if (ADD) {
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int px = 0; px < 2; ++px)
                        ad[py][px] = *reinterpret_cast<const bf16x4*>(a.addend + ((((size_t)n * a.H + 2 * m + py) * a.W + 2 * j + px) * 8 + fq * 4) * 2);
            }
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    f32x4 v = acc[py][px];
                    if (ADD) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)ad[py][px][e];
                    }
                    const bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(a.dx + ((((size_t)n * a.H + 2 * m + py) * a.W + 2 * j + px) * 8 + fq * 4) * 2) = pk;
                }
        }
    }
}

static bool s2n_dgrad_ok(const sgg_conv_desc* d) {
    return use_glds() && d->dtype == SGG_BF16 && d->pad_mode == SGG_PAD_ZERO && d->R == 3 && d->S == 3 && d->stride == 2 && d->C == 8 && d->K == 64 &&
           d->pad_t == 0 && d->pad_l == 0 && d->H == 2 * d->Ho && d->W == 2 * d->Wo;
}
static int launch_s2n_dgrad(const sgg_conv_desc* d, const void* dy, const void* w, const void* w2, int nsplit, const void* addend, void* dx, hipStream_t s) {
    S2NArgs q;
    q.dy = (const char*)dy; q.w = (const char*)w; q.w2 = (const char*)(w2 ? w2 : w); q.dx = (char*)dx; q.addend = (const char*)addend;
    q.N = d->N; q.nsplit = w2 ? nsplit : d->N; q.H = d->H; q.W = d->W; q.Ho = d->Ho; q.Wo = d->Wo;
    q.items = d->N * d->Ho * ((d->Wo + 15) / 16);
    const int waves = 256 * 16;                          // 16 waves per CU, each walking a contiguous run of items
    q.items_per_wave = (q.items + waves - 1) / waves;
    const int blocks = ((q.items + q.items_per_wave - 1) / q.items_per_wave + 3) / 4;
    if (addend) hipLaunchKernelGGL(conv3x3s2_narrow_dgrad_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, q);
    else hipLaunchKernelGGL(conv3x3s2_narrow_dgrad_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, q);
    return sgg_check_launch();
}

// -------------------------------------------------------------------------------------------------
// Weight gradient of the same layer (D.h0: x has 3 real channels, dy 64): dW[r][s][c][k] = sum_pixels x[2yo+r][2xo+s][c] dy[yo][xo][k],
// a 27 x 64 result over 262 144 pixels per 8 images.  As a GEMM the reduction runs over pixels, so both operands would have to be
// transposed through LDS for 0.45 GMAC of work; the register-staged generic kernel (v1) ran it at 20 TFLOP/s, 46.6 us incl. 21 us
// of slab reduce (now 28 + 5 us).  The work is
// small enough for the vector ALU: a lane owns one output channel k and keeps the 9 x CR sums in registers; per output pixel it
// reads its dy value (one 128-byte line per wave) and multiplies it with the pixel's 9 x CR input values, which every lane reads
// from the SAME LDS address (a broadcast read) out of three input rows staged as float32.  A block walks (image, output row, half
// row) units; its four waves split a unit's pixels, add up through LDS in fixed order at the end and write ONE slab per block.
// -------------------------------------------------------------------------------------------------
#define S2W_SEG 128                                     // output pixels per unit
#define S2W_XPIX (2 * S2W_SEG + 2)                      // input pixels per staged row (+1 used, +1 pad)
template <int CR>                                      // input channels computed (4: the first half of the padded 8)
__global__ __launch_bounds__(256) void conv3x3s2_narrow_wgrad_kernel(const char* __restrict__ x, const char* __restrict__ dy, float* __restrict__ ws,
                                                                     int N, int H, int W, int Ho, int Wo, int units, int units_per_block) {
    constexpr int CV = CR <= 4 ? 4 : 8;                 // floats per staged pixel
    // staging buffers and, after the last unit, the cross-wave reduction buffer share one allocation
    constexpr int XB = 3 * S2W_XPIX * CV * 4, DB = S2W_SEG * 64 * 2, RB = 4 * 9 * CR * 64 * 4;
    __shared__ __attribute__((aligned(16))) char sraw[(XB + DB > RB) ? (XB + DB) : RB];
    float (*sX)[S2W_XPIX][CV] = reinterpret_cast<float (*)[S2W_XPIX][CV]>(sraw);
    bf16 (*sD)[64] = reinterpret_cast<bf16 (*)[64]>(sraw + XB);
    float (*sRed)[9 * CR][64] = reinterpret_cast<float (*)[9 * CR][64]>(sraw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int segs = (Wo + S2W_SEG - 1) / S2W_SEG;
    float acc[9][CR];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < CR; ++c) acc[t][c] = 0.f;
    const int u0 = blockIdx.x * units_per_block, u1 = min(units, u0 + units_per_block);
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    // a unit's operands travel global -> registers -> LDS; the NEXT unit's loads are issued before this unit's arithmetic
    // (with the dy values read straight from global memory per pixel the kernel was latency bound: 91 us)
    constexpr int XL = (3 * S2W_XPIX + 255) / 256, DL = S2W_SEG * 8 / 256;     // 16-byte pieces per thread
    u32x4 rx[XL], rd[DL];
    auto fetch = [&](int u) {
        int b = u;
        const int seg = b % segs; b /= segs;
        const int yo = b % Ho;
        const int n = b / Ho;
        const int xo0 = seg * S2W_SEG, X0 = 2 * xo0;
#pragma unroll
        for (int l = 0; l < XL; ++l) {
            const int i = tid + 256 * l;
            const int r = i / S2W_XPIX, px = i - r * S2W_XPIX;
            const int Y = 2 * yo + r, X = X0 + px;
            // (a select between ADDRESSES, not a conditional load: that compiles to a branch with a full vmcnt(0) wait behind it)
            const bool ok = i < 3 * S2W_XPIX && Y < H && X < W;                  // bottom / right padding: zeros
            rx[l] = ld16(ok ? x + (((size_t)n * H + Y) * W + X) * 16 : zero);
        }
#pragma unroll
        for (int l = 0; l < DL; ++l) {
            const int i = tid + 256 * l, p = i >> 3, ch = i & 7;
            rd[l] = ld16(xo0 + p < Wo ? dy + ((((size_t)n * Ho + yo) * Wo + xo0 + p) * 64 + ch * 8) * 2 : zero);
        }
    };
    if (u0 < u1) fetch(u0);
    for (int u = u0; u < u1; ++u) {
        __syncthreads();                                // the previous unit's reads of sX / sD are done
#pragma unroll
        for (int l = 0; l < XL; ++l) {
            const int i = tid + 256 * l;
            if (i >= 3 * S2W_XPIX) continue;
            const int r = i / S2W_XPIX, px = i - r * S2W_XPIX;
            float v[8];
            // (opaque to the optimiser until here: it otherwise converts each piece right behind its load in fetch() -- same
            // register count -- and waits for vmcnt(0) there, four dependent round trips per unit instead of a prefetch)
            asm volatile("" : "+v"(rx[l][0]), "+v"(rx[l][1]), "+v"(rx[l][2]), "+v"(rx[l][3]));
            ET<bf16>::unpack(rx[l], v);
#pragma unroll
            for (int e = 0; e < CV; e += 4) *reinterpret_cast<f32x4*>(&sX[r][px][e]) = (f32x4){v[e], v[e + 1], v[e + 2], v[e + 3]};
        }
#pragma unroll
        for (int l = 0; l < DL; ++l) {
            const int i = tid + 256 * l;
            st16(reinterpret_cast<char*>(&sD[0][0]) + i * 16, rd[l]);
        }
        __syncthreads();
        if (u + 1 < u1) fetch(u + 1);
        // this wave's quarter of the unit, four pixels per trip
        const int p0 = wave * (S2W_SEG / 4);
        for (int p = p0; p < p0 + S2W_SEG / 4; p += 4) {
            uint16_t draw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) draw[q] = reinterpret_cast<const uint16_t*>(&sD[p + q][0])[lane];
            float dv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) dv[q] = __uint_as_float((uint32_t)draw[q] << 16);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) {
                        const float* xv = &sX[r][2 * (p + q) + sx][0];     // the same address in every lane: a broadcast read
#pragma unroll
                        for (int c = 0; c < CR; ++c) acc[r * 3 + sx][c] = fmaf(dv[q], xv[c], acc[r * 3 + sx][c]);
                    }
        }
    }
    // block total in fixed wave order -> slab [9][8][64] (rows c >= Cr are never read by the reducer)
    __syncthreads();                                    // (sRed aliases the staging buffers)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < CR; ++c) sRed[wave][t * CR + c][lane] = acc[t][c];
    __syncthreads();
    float* slab = ws + (size_t)blockIdx.x * 9 * 8 * 64;
    for (int i = tid; i < 9 * CR * 64; i += 256) {
        const int row = i >> 6, k = i & 63;
        const float v = ((sRed[0][row][k] + sRed[1][row][k]) + sRed[2][row][k]) + sRed[3][row][k];
        slab[((row / CR) * 8 + (row % CR)) * 64 + k] = v;
    }
}

// Reducer for MANY small slabs (D.h0: 512 slabs of a 27 x 64 result): wgrad_reduce_kernel gives every output its own thread, which
// then walks all slabs -- 432 threads and 64 dependent rounds of loads here.  This one puts a WAVE on each 16-byte output: lane l sums
// slabs l, l + 64, ... (all loads in flight), then a fixed-order butterfly; blockIdx.y = network of a two-network launch.
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* ws, float* dw, float* dw2, int taps, int C, int K, int Cr, int Kr, int splits, int accumulate) {
    const int K4 = K / 4, lane = threadIdx.x & 63;
    const int total = taps * Cr * K4;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= total) return;
    const int net = blockIdx.y;
    float* const out = net ? dw2 : dw;
    const size_t slab = (size_t)taps * C * K;
    const int k4 = i % K4, t = i / K4, c = t % Cr, tap = t / Cr;
    const float* src = ws + (size_t)net * splits * slab + ((size_t)tap * C + c) * K + k4 * 4;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int sp = lane; sp < splits; sp += 64) v += *reinterpret_cast<const f32x4*>(src + (size_t)sp * slab);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], off);
    if (lane == 0) {
        float* o = out + ((size_t)tap * Cr + c) * Kr + k4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k4 * 4 + e < Kr) o[e] = accumulate ? o[e] + v[e] : v[e];
    }
}

#define S2W_BLOCKS 512                                  // persistent blocks = slabs (1024 / 2048 measured 7 / 22 % slower)
static bool s2n_wgrad_ok(const sgg_conv_desc* d) {
    return use_glds() && d->dtype == SGG_BF16 && d->pad_mode == SGG_PAD_ZERO && d->R == 3 && d->S == 3 && d->stride == 2 && d->C == 8 && d->K == 64 &&
           d->pad_t == 0 && d->pad_l == 0 && d->H == 2 * d->Ho && d->W == 2 * d->Wo;
}
static int s2n_wgrad_blocks(const sgg_conv_desc* d) {
    const int units = d->N * d->Ho * ((d->Wo + S2W_SEG - 1) / S2W_SEG);
    const int upb = (units + S2W_BLOCKS - 1) / S2W_BLOCKS;
    return (units + upb - 1) / upb;
}
static int launch_s2n_wgrad(const sgg_conv_desc* d, const void* x, const void* dy, float* ws, int Cr, hipStream_t s) {
    const int units = d->N * d->Ho * ((d->Wo + S2W_SEG - 1) / S2W_SEG);
    const int upb = (units + S2W_BLOCKS - 1) / S2W_BLOCKS, blocks = (units + upb - 1) / upb;
    if (Cr > 4) return SGG_EUNSUPPORTED;                  // (run_wgrad sends wider inputs to the generic kernel)
    // four channels per pixel although D.h0 has three (the fourth is the tensor's zero padding): 36 sums = 18 register PAIRS
    // for v_pk_fma_f32 and one ds_read_b128 per tap; with 27 sums hipcc spent 90 v_mov per four pixels lining operands up
    hipLaunchKernelGGL(conv3x3s2_narrow_wgrad_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, (const char*)x, (const char*)dy, ws,
                       d->N, d->H, d->W, d->Ho, d->Wo, units, upb);
    return sgg_check_launch();
}

static bool n7_fwd_ok(const sgg_conv_desc* d) {
    return sgg_config().n7 && d->dtype == SGG_BF16 && d->R == 7 && d->S == 7 && d->stride == 1 && d->C == 64 && d->K == 8 &&
           d->Ho == d->H && d->Wo == d->W && d->pad_t == 3 && d->pad_l == 3 && d->H >= 8 && d->W >= 8;
}
// stem data gradient: conv 7x7 REFLECT-3, <= 3 -> 64 channels; as a conv over dy it is the head's shape with mirrored taps
static bool n7_dgrad_ok(const sgg_conv_desc* d) {
    return sgg_config().n7 && d->dtype == SGG_BF16 && d->R == 7 && d->S == 7 && d->stride == 1 && d->K == 64 && d->C == 8 &&
           d->Ho == d->H && d->Wo == d->W && d->pad_t == 3 && d->pad_l == 3 && d->pad_mode == SGG_PAD_REFLECT && d->H >= 8 && d->W >= 8;
}
static size_t n7_dgrad_ws(const sgg_conv_desc* d) { return (size_t)d->N * (d->H + 6) * (d->W + 6) * 16; }

static int launch_n7(const N7Args& a, hipStream_t s) {
    SGG_LDS_ATTR(conv7_narrow_out_kernel, N7_LDS);
    const int64_t tiles = (int64_t)a.N * ((a.Ho + N7_TH - 1) / N7_TH) * ((a.Wo + N7_TW - 1) / N7_TW);
    const int64_t per = (tiles + 255) / 256;            // persistent blocks, one per CU, equal tile counts
    const int64_t blocks = (tiles + per - 1) / per;
    hipLaunchKernelGGL(conv7_narrow_out_kernel, dim3((unsigned)blocks), dim3(512), N7_LDS, s, a);
    return sgg_check_launch();
}

static bool halo_fwd_ok(const sgg_conv_desc* d) {
    const int cch = d->dtype == SGG_BF16 ? 64 : 32;
    return d->K <= 16 && d->stride == 1 && d->R <= HALO_MAXR && d->S <= HALO_MAXR && d->Ho == d->H && d->Wo == d->W &&
           d->H % HALO_TH == 0 && d->W % HALO_TW == 0 && d->C % cch == 0;
}

template <typename T>
static int launch_halo_fwd(const sgg_conv_desc* d, const ConvArgs& a, hipStream_t s, int flip = 0) {
    size_t lds = 2 * (size_t)HALO_MAXR * 16 * 128 + (size_t)(HALO_TH + d->R - 1) * (HALO_TW + d->S - 1) * 128 + 1024;
    auto kern = conv_halo_fwd_kernel<T>;
    SGG_LDS_ATTR(kern, 160 * 1024);
    dim3 grid((unsigned)(d->N * (d->H / HALO_TH) * (d->W / HALO_TW)));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a, flip);
    return sgg_check_launch();
}

// data gradient of a narrow-INPUT conv (the stem, 3->64 7x7; cycle mode back-propagates into the fake image): as a
// convolution over dy it is the head's shape (64 -> <= 16 channels), so it runs conv_halo_fwd_kernel with mirrored taps
// and zero padding; REFLECT's MirrorPadGrad terms are then added to the border pixels by the MODE_BORDER launch
static bool halo_dgrad_narrow_ok(const sgg_conv_desc* d) {
    const int en = sgg_config().stem_dgrad_halo;
    const int cch = d->dtype == SGG_BF16 ? 64 : 32;
    return en && use_glds() && d->C <= 16 && d->stride == 1 && d->R <= HALO_MAXR && d->S <= HALO_MAXR && d->R == d->S &&
           d->Ho == d->H && d->Wo == d->W && d->pad_t == (d->R - 1) / 2 && d->pad_l == (d->S - 1) / 2 &&
           d->H % HALO_TH == 0 && d->W % HALO_TW == 0 && d->K % cch == 0;
}

// Weight gradient of the same narrow-output convolution: dW[(r,s,c)][k] = sum_pixels x~[p,(r,s),c] * dy[p][k] with
// k <= 16, c = 64.  Blocks walk 16x32-pixel tiles persistently; the input halo (DMA) and the dy tile sit in LDS;
// wave w owns taps w, w+8, ... (<= 7 of the 49) for all 64 channels, accumulating in registers across its tiles;
// both MFMA operands come from transposing LDS reads (pixels are the reduction index).  One f32 slab per block,
// summed by wgrad_reduce_kernel.
template <typename T>
__global__ __launch_bounds__(512) void conv_halo_wgrad_kernel(WgradArgs a, int ntiles) {
    constexpr int ES = (int)sizeof(T);
    constexpr int CCH = 128 / ES;                       // channels per halo pass
    constexpr int CF = CCH / 16;                        // 16-channel fragments per pass (4 bf16 / 2 f32)
    constexpr int DPITCH = 16 * ES;                     // dy tile: 16 couts per pixel (zero padded)
    constexpr int TSLOTS = 7;                           // taps per wave (R*S <= 56)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sD = smem;                                    // [512 px][DPITCH]
    char* sH = smem + 512 * DPITCH;                     // halo

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index as a scalar
    const int HWd = HALO_TW + a.S - 1, HHd = HALO_TH + a.R - 1, HP = HHd * HWd;
    const int tilesW = a.W / HALO_TW, tilesH = a.H / HALO_TH;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    const int pos = tid & 7, lsw = (tid >> 4) & 7, lcc = pos ^ lsw;
    const int ntaps = a.R * a.S;
    const int g = lane >> 4, u = lane & 15, q = u >> 2, pp = u & 3;

    f32x4 acc[TSLOTS][4];
#pragma unroll
    for (int i = 0; i < TSLOTS; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int i = tid; i < 512 * DPITCH / 16; i += 512) st16(sD + i * 16, zero16());   // padded couts stay zero
    const int nchunks = a.C / CCH;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b = tile;
        const int tw = b % tilesW; b /= tilesW;
        const int th = b % tilesH;
        const int n = b / tilesH;
        const int y0 = th * HALO_TH, x0 = tw * HALO_TW;
        for (int cc = 0; cc < nchunks; ++cc) {
            __syncthreads();
            for (int base = 0; base < HP; base += 64) {
                int hp = base + (tid >> 3);
                const char* src = zero;
                if (hp < HP) {
                    int hy = hp / HWd, hx = hp - hy * HWd;
                    int yi = y0 - a.pad_t + hy, xi = x0 - a.pad_l + hx;
                    bool ok = true;
                    if (a.reflect) {
                        yi = yi < 0 ? -yi : (yi >= a.H ? 2 * (a.H - 1) - yi : yi);
                        xi = xi < 0 ? -xi : (xi >= a.W ? 2 * (a.W - 1) - xi : xi);
                    } else ok = (unsigned)yi < (unsigned)a.H && (unsigned)xi < (unsigned)a.W;
                    if (ok) src = a.x + ((((size_t)n * a.H + yi) * a.W + xi) * a.C + cc * CCH) * ES + lcc * 16;
                }
                if (base + wave * 8 < HP)
                    dma16_to_lds(src, (__attribute__((address_space(3))) void*)(sH + (base + wave * 8) * 128));
            }
            if (cc == 0) {                               // dy tile: thread = pixel; K*ES bytes of real data per pixel
                const int py = tid >> 5, px = tid & 31;
                const char* src = a.dy + (((size_t)n * a.H + y0 + py) * a.W + x0 + px) * a.K * ES;
                for (int o = 0; o < a.K * ES; o += 16) st16(sD + tid * DPITCH + o, ld16(src + o));
            }
            SGG_WAIT_VM0();
            __syncthreads();
            for (int y = 0; y < HALO_TH; ++y) {           // one k-step = the 32 pixels of tile row y
                if constexpr (sizeof(T) == 2) {
                    const int p0 = y * 32 + 8 * g + q;
                    bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sD + p0 * DPITCH + pp * 8));
                    bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sD + (p0 + 4) * DPITCH + pp * 8));
                    const bf16x8 fb = (bf16x8){blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
#pragma unroll
                    for (int i = 0; i < TSLOTS; ++i) {
                        const int t = wave + 8 * i;
                        if (t >= ntaps) break;
                        const int r = t / a.S, s = t - r * a.S;
                        const int h0 = (y + r) * HWd + s + 8 * g + q, h1 = h0 + 4;
#pragma unroll
                        for (int c = 0; c < CF; ++c) {
                            const int col = c * 16 + 4 * pp, ch = col >> 3, sub = (col & 7) * 2;
                            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                (__attribute__((address_space(3))) bf16x4*)(sH + h0 * 128 + ((ch ^ ((h0 >> 1) & 7)) << 4) + sub));
                            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                (__attribute__((address_space(3))) bf16x4*)(sH + h1 * 128 + ((ch ^ ((h1 >> 1) & 7)) << 4) + sub));
                            const bf16x8 fa = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i][c], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int s4 = 0; s4 < 8; ++s4) {          // 4 pixels per MFMA
                        const int xk = 4 * s4 + g;
                        const float fb = *reinterpret_cast<const float*>(sD + (y * 32 + xk) * DPITCH + u * 4);
#pragma unroll
                        for (int i = 0; i < TSLOTS; ++i) {
                            const int t = wave + 8 * i;
                            if (t >= ntaps) break;
                            const int r = t / a.S, s = t - r * a.S;
                            const int hp = (y + r) * HWd + s + xk;
#pragma unroll
                            for (int c = 0; c < CF; ++c) {
                                const int col = c * 16 + u;
                                const float fa = *reinterpret_cast<const float*>(sH + hp * 128 + (((col >> 2) ^ ((hp >> 1) & 7)) << 4) + (col & 3) * 4);
                                acc[i][cc * CF + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[i][cc * CF + c], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    }

    // D[row = 4g+e -> channel][col = u -> cout]
    float* slab = a.ws + (size_t)blockIdx.x * ntaps * a.C * a.K;
    if (u < a.K) {
#pragma unroll
        for (int i = 0; i < TSLOTS; ++i) {
            const int t = wave + 8 * i;
            if (t >= ntaps) break;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c * 16 >= a.C) break;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    slab[((size_t)t * a.C + c * 16 + g * 4 + e) * a.K + u] = acc[i][c][e];
            }
        }
    }
}

static bool halo_wgrad_ok(const sgg_conv_desc* d) {
    const int cch = d->dtype == SGG_BF16 ? 64 : 32;
    return use_glds() && d->K <= 16 && d->stride == 1 && d->R <= HALO_MAXR && d->S <= HALO_MAXR && d->R * d->S <= 56 &&
           d->Ho == d->H && d->Wo == d->W && d->H % HALO_TH == 0 && d->W % HALO_TW == 0 && d->C <= 64 && d->C % cch == 0;
}
static int halo_wgrad_blocks(const sgg_conv_desc* d) {
    int ntiles = d->N * (d->H / HALO_TH) * (d->W / HALO_TW);
    return ntiles < 256 ? ntiles : 256;
}

// Narrow-INPUT convolution (8 source channels = one 16-byte chunk per pixel, 64 output channels, stride 1, same-size):
// the generator stem 7x7 3(8)->64 forward (module.py:230-232) and the data-gradient of the head 7x7 64->3(8).  As an
// implicit GEMM every (pixel, tap) is a separate 16-byte DMA element; here the source halo of a 16x32 tile (13 KB) and
// the whole weight matrix (64 x 49*8, 53 KB) sit in LDS, each lane picks the chunk of ITS tap out of the halo, and
// persistent blocks amortise the weight load.  flip = data-gradient (taps mirrored, zero fill).
#ifndef NI_WIDE
#define NI_WIDE 1                                      // bf16 epilogue with 16-byte stores (+ the optional norm-statistics rows)
#endif
template <typename T>
__global__ __launch_bounds__(512) void conv_halo_narrow_in_kernel(ConvArgs a, int ntiles, int flip) {
    constexpr int VEC = ET<T>::VEC;
    constexpr int ES = (int)sizeof(T);
    constexpr int SCH = 8;                              // source channels
    constexpr int CPV = SCH / VEC;                      // 16-byte chunks per (pixel, tap): 1 bf16, 2 f32
    constexpr int PXB = SCH * ES;                       // bytes per halo pixel
    constexpr int DCH = 64;                             // destination channels
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int taps = a.R * a.S;
    const int kch = taps * CPV;                         // real 16-byte chunks per weight row
    const int ksteps = (kch + 3) / 4;
    const int wpitch = ((ksteps * 4) | 1) * 16;         // odd chunk count per row: conflict-free fragment reads
    char* sW = smem;                                    // [64][wpitch]
    char* sH = smem + DCH * wpitch;                     // halo [(TH+R-1)*(TW+S-1)][PXB]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index as a scalar
    const int HWd = HALO_TW + a.S - 1, HHd = HALO_TH + a.R - 1, HP = HHd * HWd;
    const int tilesW = a.W / HALO_TW, tilesH = a.H / HALO_TH;
    const int frow = lane & 15, fq = lane >> 4;

    // weights: rows = destination channels, zero beyond the real K
    for (int idx = tid; idx < DCH * ksteps * 4; idx += 512) {
        int row = idx / (ksteps * 4), ch = idx - row * (ksteps * 4);
        u32x4 v = zero16();
        if (ch < kch) v = ld16(a.wmat + ((size_t)row * taps * SCH) * ES + ch * 16);
        st16(sW + row * wpitch + ch * 16, v);
    }
    // origin of the source halo relative to the tile: fwd reads (y - p + r), the data-gradient (y + p - r)
    const int oy = flip ? a.pad_t - (a.R - 1) : -a.pad_t, ox = flip ? a.pad_l - (a.S - 1) : -a.pad_l;
    // per 16-byte k-chunk q: byte offset of its tap inside the halo, ((tr * HWd + ts) * PXB + sub * 16); chunks past the last
    // tap read tap 0 (their weights are zero)
    int* sTap = reinterpret_cast<int*>(sH + (size_t)HP * PXB);
    float* sS = reinterpret_cast<float*>(sTap + ksteps * 4);      // [8 waves][64 channels][2]: norm-statistics epilogue (a.stats)
    for (int q = tid; q < ksteps * 4; q += 512) {
        int tap = q / CPV, sub = q - tap * CPV;
        if (tap >= taps) { tap = 0; sub = 0; }
        int tr = tap / a.S, ts = tap - tr * a.S;
        if (flip) { tr = a.R - 1 - tr; ts = a.S - 1 - ts; }
        sTap[q] = (tr * HWd + ts) * PXB + sub * 16;
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b = tile;
        const int tw = b % tilesW; b /= tilesW;
        const int th = b % tilesH;
        const int n = b / tilesH;
        const int y0 = th * HALO_TH, x0 = tw * HALO_TW;
        __syncthreads();                                // previous tile's halo reads are done; weights are visible
        for (int idx = tid; idx < HP * CPV; idx += 512) {
            int hp = idx / CPV, sub = idx - hp * CPV;
            int hy = hp / HWd, hx = hp - hy * HWd;
            int yi = y0 + oy + hy, xi = x0 + ox + hx;
            bool ok = true;
            if (a.reflect && !flip) {
                yi = yi < 0 ? -yi : (yi >= a.H ? 2 * (a.H - 1) - yi : yi);
                xi = xi < 0 ? -xi : (xi >= a.W ? 2 * (a.W - 1) - xi : xi);
            } else ok = (unsigned)yi < (unsigned)a.H && (unsigned)xi < (unsigned)a.W;
            u32x4 v = zero16();
            if (ok) v = ld16(a.src + (((size_t)n * a.H + yi) * a.W + xi) * PXB + sub * 16);
            st16(sH + hp * PXB + sub * 16, v);
        }
        __syncthreads();

        f32x4 acc[4][4];                                // [cout frag][pixel frag]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < ksteps; ++ks) {
            const int qq = ks * 4 + fq;                  // this lane's 16-byte k-chunk
            // halo byte offset of this lane's tap (tr, ts, sub): a table built once per block (the two integer divisions and the
            // flip per k-step and lane made this loop VALU-issue bound: PMC showed the matrix pipes 34 % busy, 44 % of wave cycles
            // issue-stalled)
            const int toff = sTap[qq];
            u32x4 fw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fw[i] = ld16(sW + (i * 16 + frow) * wpitch + qq * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = 2 * wave + (j >> 1), x = (j & 1) * 16 + frow;
                u32x4 fp = ld16(sH + (y * HWd + x) * PXB + toff);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (sizeof(T) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[i]), __builtin_bit_cast(bf16x8, fp), acc[i][j], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw[i][e]), __uint_as_float(fp[e]), acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        float bv[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[i][e] = a.bias ? a.bias[i * 16 + fq * 4 + e] : 0.f;
        act_dispatch(a.act, [&](auto act_c) {
            constexpr int ACT = decltype(act_c)::value;
            if constexpr (sizeof(T) == 2 && NI_WIDE) {
                // 16-byte stores (the two 16-column fragments of a tile row trade halves between lane rows, see conv3x3_halo_gemm_kernel) and,
                // with a.stats, the (sum, sumsq) rows of the stored output for the instance norm behind the stem (module.py:230-233):
                // wave butterfly -> the eight waves' rows through LDS -> ONE partial row per (image, 16 x 32 tile)
                auto ep = [&](auto st_c) {
                    constexpr bool ST = decltype(st_c)::value;
                    const int jo = fq & 1, cb = (fq >> 1) * 8;
                    float s1[4][4], s2[4][4];
                    if constexpr (ST) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int e = 0; e < 4; ++e) s1[i][e] = s2[i][e] = 0.f;
                    }
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        const int y = y0 + 2 * wave + jp, x = x0 + jo * 16 + frow;
                        const size_t dpix = ((size_t)n * a.H + y) * a.W + x;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float v0[4], v1[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v0[e] = act_apply_c<ACT>(acc[i][2 * jp][e] + bv[i][e], a.leak);
                                v1[e] = act_apply_c<ACT>(acc[i][2 * jp + 1][e] + bv[i][e], a.leak);
                            }
                            const bf16x4 p0 = {(bf16)v0[0], (bf16)v0[1], (bf16)v0[2], (bf16)v0[3]}, p1 = {(bf16)v1[0], (bf16)v1[1], (bf16)v1[2], (bf16)v1[3]};
                            if constexpr (ST) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { const float r0 = (float)p0[e]; s1[i][e] += r0; s2[i][e] += r0 * r0; }
#pragma unroll
                                for (int e = 0; e < 4; ++e) { const float r1 = (float)p1[e]; s1[i][e] += r1; s2[i][e] += r1 * r1; }
                            }
                            const u32x2 q0 = __builtin_bit_cast(u32x2, p0), q1 = __builtin_bit_cast(u32x2, p1);
                            uint32_t a0 = q0[0], a1 = q0[1], b0 = q1[0], b1 = q1[1];
                            row_swap16(a0, b0); row_swap16(a1, b1);
                            st16(reinterpret_cast<bf16*>(a.dst) + dpix * DCH + i * 16 + cb, (u32x4){a0, a1, b0, b1});
                        }
                    }
                    if constexpr (ST) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
#pragma unroll
                                for (int off = 1; off < 16; off <<= 1) {
                                    s1[i][e] += __shfl_xor(s1[i][e], off);
                                    s2[i][e] += __shfl_xor(s2[i][e], off);
                                }
                            }
                        if (frow == 0) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float* o = sS + (wave * 64 + i * 16 + fq * 4 + e) * 2;
                                    o[0] = s1[i][e]; o[1] = s2[i][e];
                                }
                        }
                        __syncthreads();
                        if (tid < 128) {
                            const int ch = tid >> 1, which = tid & 1;
                            float t = 0.f;
#pragma unroll
                            for (int w8 = 0; w8 < 8; ++w8) t += sS[(w8 * 64 + ch) * 2 + which];
                            a.stats[(((size_t)n * (tilesH * tilesW) + th * tilesW + tw) * DCH + ch) * 2 + which] = t;
                        }
                        // (the next tile's rows are written behind the two barriers at the head of its loop iteration)
                    }
                };
                if (a.stats) ep(std::true_type{}); else ep(std::false_type{});
                return;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = y0 + 2 * wave + (j >> 1), x = x0 + (j & 1) * 16 + frow;
                const size_t dpix = ((size_t)n * a.H + y) * a.W + x;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int dc = i * 16 + fq * 4;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_apply_c<ACT>(acc[i][j][e] + bv[i][e], a.leak);
                    T* o = reinterpret_cast<T*>(a.dst) + dpix * DCH + dc;
                    if constexpr (sizeof(T) == 2) {
                        bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                        *reinterpret_cast<bf16x4*>(o) = pk;
                    } else *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
                }
            }
        });
    }
}

// src_ch / dst_ch: channel counts of the gathered tensor and of the result for the direction in question
static bool halo_narrow_in_ok(const sgg_conv_desc* d, int src_ch, int dst_ch) {
    return use_glds() && src_ch == 8 && dst_ch == 64 && d->stride == 1 && d->R <= HALO_MAXR && d->S <= HALO_MAXR &&
           d->Ho == d->H && d->Wo == d->W && d->H % HALO_TH == 0 && d->W % HALO_TW == 0;
}

template <typename T>
static int launch_halo_narrow_in(const sgg_conv_desc* d, const ConvArgs& a, int flip, hipStream_t s) {
    const int cpv = 8 / (16 / (int)sizeof(T));
    const int ksteps = (d->R * d->S * cpv + 3) / 4;
    size_t lds = (size_t)64 * (((ksteps * 4) | 1) * 16) + (size_t)(HALO_TH + d->R - 1) * (HALO_TW + d->S - 1) * 8 * sizeof(T)
                 + (size_t)ksteps * 4 * sizeof(int)                        // + the tap-offset table
                 + (size_t)8 * 64 * 2 * sizeof(float);                     // + the statistics epilogue's per-wave rows
    auto kern = conv_halo_narrow_in_kernel<T>;
    SGG_LDS_ATTR(kern, 160 * 1024);
    int ntiles = d->N * (d->H / HALO_TH) * (d->W / HALO_TW);
    int blocks = ntiles < 512 ? ntiles : 512;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, s, a, ntiles, flip);
    return sgg_check_launch();
}

// Side tensor for the REFLECT data-gradient: for every border pixel (the pixels MirrorPadGrad adds mirrored terms
// to) and every tap, the gather row the GEMM needs = sum of dy over all preimages of the pixel's padded position.
// fold[b][tap][k], b enumerating border pixels per image: the 2p border ROW bands first (all columns), then the
// 2p border COLUMNS of the remaining rows (or every pixel when the bands overlap).
__host__ __device__ inline int fold_border_per_image(int H, int W, int p) {
    int p2 = 2 * p;
    return (H <= p2 + 1 || W <= p2 + 1) ? H * W : p2 * W + (H - p2) * p2;
}

template <typename T>
__global__ __launch_bounds__(256) void fold_gather_kernel(const char* dy, char* fold, int N, int H, int W, int K, int R, int S, int p, int Ho, int Wo) {
    constexpr int VEC = ET<T>::VEC;
    const int cpv = K / VEC, taps = R * S, p2 = 2 * p;
    const bool all = H <= p2 + 1 || W <= p2 + 1;
    const int Bimg = fold_border_per_image(H, W, p);
    const int64_t total = (int64_t)N * Bimg * taps * cpv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int ch = (int)(i % cpv);
        int64_t t = i / cpv;
        int tap = (int)(t % taps);
        int b = (int)(t / taps);
        int n = b / Bimg, q = b - n * Bimg, h, w;
        if (all) { h = q / W; w = q - h * W; }
        else if (q < p2 * W) { int hi = q / W; w = q - hi * W; h = hi < p ? 1 + hi : H - 1 - p + (hi - p); }
        else {
            q -= p2 * W;
            int hh = q / p2, wi = q - hh * p2;
            h = hh == 0 ? 0 : (hh == H - p2 - 1 ? H - 1 : p + hh);
            w = wi < p ? 1 + wi : W - 1 - p + (wi - p);
        }
        int r = tap / S, s = tap - r * S;
        int jh[3], jw[3];
        int nh = reflect_preimages(h, H, p, jh), nw = reflect_preimages(w, W, p, jw);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int ia = 0; ia < nh; ++ia)
            for (int ib = 0; ib < nw; ++ib) {
                int ho = jh[ia] - r, wo = jw[ib] - s;
                if ((unsigned)ho < (unsigned)Ho && (unsigned)wo < (unsigned)Wo) {
                    float v[VEC];
                    ET<T>::unpack(ld16(dy + ((((size_t)n * Ho + ho) * Wo + wo) * K + (size_t)ch * VEC) * sizeof(T)), v);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += v[e];
                }
            }
        st16(fold + (size_t)i * 16, ET<T>::pack(acc));
    }
}

// -------------------------------------------------------------------------------------------------
// weight gradient
// -------------------------------------------------------------------------------------------------

__device__ inline int wg_swz(int chunk, int row, int nchunks) {
    int f = (row & 3) | (((row >> 3) & 1) << 2);
    return chunk ^ ((f << 1) & (nchunks - 1));
}

template <typename T, int BMW, int BNW, int WGM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int VEC = ET<T>::VEC;
    constexpr int ES = (int)sizeof(T);
    constexpr int BKP = 32;                               // pixels per K-step
    constexpr int WGN = 4 / WGM;
    constexpr int WMW = BMW / WGM, WNW = BNW / WGN;
    constexpr int MI = WMW / 16, NI = WNW / 16;
    constexpr int NCX = BMW / VEC, NCD = BNW / VEC;       // 16-byte chunks per tile row
    constexpr int RPX = 256 / NCX, RPD = 256 / NCD;       // pixel rows covered per pass
    constexpr int PX = (BKP + RPX - 1) / RPX, PD = (BKP + RPD - 1) / RPD;
    constexpr int XP = BMW * ES, DP = BNW * ES;           // row pitch (bytes)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;                       // [2][BKP][XP]
    char* sD = smem + 2 * BKP * XP;        // [2][BKP][DP]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index as a scalar
    const int wm = wave / WGN, wn = wave % WGN;
    const int Mrows = a.R * a.S * a.C;
    const int m0 = blockIdx.x * BMW, n0 = blockIdx.y * BNW;
    const int pbeg = blockIdx.z * a.pix_per_split;
    const int pend = min(a.P, pbeg + a.pix_per_split);

    // fixed chunk column of this thread in each tile
    const int jx = tid % NCX, rx = tid / NCX;
    const int jd = tid % NCD, rd = tid / NCD;
    const int mrow = m0 + jx * VEC;
    const bool xcol_ok = mrow < Mrows;
    const int tap = xcol_ok ? mrow / a.C : 0, c0 = mrow - tap * a.C;
    const int tr = tap / a.S, ts = tap - tr * a.S;
    const bool dcol_ok = (n0 + jd * VEC) < a.K;

    u32x4 regX[PX], regD[PD];
    auto load_tile = [&](int p0) {
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            int prow = rx + i * RPX, p = p0 + prow;
            regX[i] = zero16();
            if (prow < BKP && xcol_ok && p < pend) {
                uint32_t n = fdiv((uint32_t)p, a.dHW), rem = (uint32_t)p - n * a.dHW.d;
                uint32_t ho = fdiv(rem, a.dW), wo = rem - ho * a.dW.d;
                int hi = (int)ho * a.stride - a.pad_t + tr, wi = (int)wo * a.stride - a.pad_l + ts;
                bool ok = true;
                if (a.reflect) {
                    hi = hi < 0 ? -hi : (hi >= a.H ? 2 * (a.H - 1) - hi : hi);
                    wi = wi < 0 ? -wi : (wi >= a.W ? 2 * (a.W - 1) - wi : wi);
                } else ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                if (ok) regX[i] = ld16(a.x + ((((size_t)n * a.H + hi) * a.W + wi) * a.C + c0) * ES);
            }
        }
#pragma unroll
        for (int i = 0; i < PD; ++i) {
            int prow = rd + i * RPD, p = p0 + prow;
            regD[i] = zero16();
            if (prow < BKP && dcol_ok && p < pend)
                regD[i] = ld16(a.dy + ((size_t)p * a.K + n0 + jd * VEC) * ES);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            int prow = rx + i * RPX;
            if (prow < BKP) st16(sX + (buf * BKP + prow) * XP + (wg_swz(jx, prow, NCX) << 4), regX[i]);
        }
#pragma unroll
        for (int i = 0; i < PD; ++i) {
            int prow = rd + i * RPD;
            if (prow < BKP) st16(sD + (buf * BKP + prow) * DP + (wg_swz(jd, prow, NCD) << 4), regD[i]);
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, u = lane & 15;
    const int ksteps = pend > pbeg ? (pend - pbeg + BKP - 1) / BKP : 0;
    if (ksteps > 0) { load_tile(pbeg); store_tile(0); }
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < ksteps) load_tile(pbeg + (ks + 1) * BKP);
        const char* bX = sX + buf * BKP * XP;
        const char* bD = sD + buf * BKP * DP;
        if constexpr (sizeof(T) == 2) {
            // transposing LDS read: 16-lane group g fetches pixels 8g..8g+7 (two 4-row blocks) x 16 columns;
            // lane u supplies row (u>>2), columns 4*(u&3)..+3 and receives column u of the 4 rows.
            const int q = u >> 2, pp = u & 3;
            bf16x8 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int col = wm * WMW + i * 16 + 4 * pp;          // element column in the X tile
                int ch = col >> 3, sub = (col & 7) * 2;
                int r0 = 8 * g + q, r1 = r0 + 4;
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(bX + r0 * XP + (wg_swz(ch, r0, NCX) << 4) + sub));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(bX + r1 * XP + (wg_swz(ch, r1, NCX) << 4) + sub));
                fa[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int col = wn * WNW + j * 16 + 4 * pp;
                int ch = col >> 3, sub = (col & 7) * 2;
                int r0 = 8 * g + q, r1 = r0 + 4;
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(bD + r0 * DP + (wg_swz(ch, r0, NCD) << 4) + sub));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(bD + r1 * DP + (wg_swz(ch, r1, NCD) << 4) + sub));
                fb[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int s4 = 0; s4 < BKP / 4; ++s4) {
                int row = 4 * s4 + g;
                float fa[MI], fb[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    int col = wm * WMW + i * 16 + u;
                    fa[i] = *reinterpret_cast<const float*>(bX + row * XP + (wg_swz(col >> 2, row, NCX) << 4) + (col & 3) * 4);
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    int col = wn * WNW + j * 16 + u;
                    fb[j] = *reinterpret_cast<const float*>(bD + row * DP + (wg_swz(col >> 2, row, NCD) << 4) + (col & 3) * 4);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        if (ks + 1 < ksteps) store_tile(buf ^ 1);
        __syncthreads();
    }

    // D[row = 4g+e -> (tap,c)][col = u -> k]
    float* slab = a.ws + (size_t)blockIdx.z * Mrows * a.K;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            int k = n0 + wn * WNW + j * 16 + u;
            if (k >= a.K) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int mr = m0 + wm * WMW + i * 16 + g * 4 + e;
                if (mr < Mrows) slab[(size_t)mr * a.K + k] = acc[i][j][e];
            }
        }
}

// -------------------------------------------------------------------------------------------------
// v2 weight gradient: 256 (tap,c rows) x 256 (couts) tiles, 8 waves, direct-to-LDS double-buffered staging.
// Same maths as conv_wgrad_kernel; the pixel-major tiles land in LDS by DMA (1 KiB per wave-instruction = whole
// tile rows), the XOR swizzle that makes the transposing reads conflict-free is applied on the source side.
// -------------------------------------------------------------------------------------------------
template <typename T> __device__ inline int wg2_key(int row) {
    if constexpr (sizeof(T) == 2) return ((row & 3) | (((row >> 3) & 1) << 2)) << 1;
    else return (row & 3) << 1;
}

// ROWFAST (Wo % BKP == 0): every K-stage lies inside one output row, so image / row / validity of the gathered x row are
// block-uniform (scalar) and a lane only resolves its column; the general path divides per lane and per pass, which
// cost ~20 us of 125 at the residual-block shape (all 8 waves do address arithmetic at the same time, MFMA pipe idle).
template <typename T, bool ROWFAST>
__global__ __launch_bounds__(512) void conv_wgrad_glds_kernel(WgradArgs a) {
    constexpr int VEC = ET<T>::VEC;
    constexpr int ES = (int)sizeof(T);
    constexpr int BT = 256;                               // tile rows (tap,c) and columns (couts)
    constexpr int BKP = sizeof(T) == 2 ? 64 : 32;         // pixels per stage
    constexpr int PITCH = BT * ES;                        // tile row pitch in bytes (512 / 1024)
    constexpr int NCH = PITCH / 16;                       // 16-byte chunks per tile row (32 / 64)
    constexpr int RPS = 512 / NCH;                        // tile rows staged per pass (16 / 8)
    constexpr int NP = BKP / RPS;                         // passes per tile (4)
    constexpr int TILE = BKP * PITCH;                     // bytes per operand tile (32 KB)
    constexpr int STAGE = 2 * TILE;
    constexpr int MI = 8, NI = 4;                         // wave tile 128 (rows) x 64 (couts); waves 2 x 4

    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages of { X [BKP][PITCH], DY [BKP][PITCH] }

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index as a scalar
    const int wm = wave >> 2, wn = wave & 3;
    const int Mrows = a.R * a.S * a.C;
    // 1-D grid, XCD-aware: blocks b, b+8, ... share an XCD (and its L2); give each XCD a contiguous range of logical ids
    // with the (tap-row tile, cout tile) index fastest, so all tiles that stream the SAME pixel range run on ONE XCD and
    // x / dy are fetched from HBM once per split instead of once per tile (PMC: 14 % -> L2 hits; speed only).
    const int tilesM = (Mrows + BT - 1) / BT, tilesN = (a.K + BT - 1) / BT, tiles = tilesM * tilesN;
    int lid;
    {
        const int b = (int)blockIdx.x, nm = (int)gridDim.x;
        const int q = nm >> 3, rr = nm & 7, xcd = b & 7;
        lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
    }
    const int split = lid / tiles, tl = lid - split * tiles;
    const int m0 = (tl / tilesN) * BT, n0 = (tl % tilesN) * BT;
    const bool netb = split >= a.splits_per_net;          // second network's blocks (uniform per block)
    if (netb) { a.x = a.xb; a.dy = a.dyb; }
    const int pbeg = (netb ? split - a.splits_per_net : split) * a.pix_per_split;
    const int pend = min(a.P, pbeg + a.pix_per_split);

    // this thread stages LDS position `pos` of rows prow0 + RPS*i; it holds logical chunk pos ^ key(row)
    const int pos = tid % NCH, prow0 = tid / NCH;
    const int lc = pos ^ (wg2_key<T>(prow0) & (NCH - 1));            // key(row) is the same for every pass (see wg2_key)
    const int mrow = m0 + lc * VEC;
    const bool xcol_ok = mrow < Mrows;
    const int tap = xcol_ok ? mrow / a.C : 0, c0 = mrow - tap * a.C;
    const int tr = tap / a.S, ts = tap - tr * a.S;
    const bool dcol_ok = (n0 + lc * VEC) < a.K;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    auto stage_tile = [&](int stg, int p0) {
        char* sX = smem + stg * STAGE;
        char* sD = sX + TILE;
        if constexpr (ROWFAST) {
            // p0 is uniform and a multiple of BKP; P and the split size are multiples of BKP too: no tail, no row crossing
            const uint32_t n = fdiv((uint32_t)p0, a.dHW), rem = (uint32_t)p0 - n * a.dHW.d;
            const uint32_t ho = fdiv(rem, a.dW), wo0 = rem - ho * a.dW.d;
            int hi = (int)ho * a.stride - a.pad_t + tr;
            bool rowok = true;
            if (a.reflect) hi = hi < 0 ? -hi : (hi >= a.H ? 2 * (a.H - 1) - hi : hi);
            else rowok = (unsigned)hi < (unsigned)a.H;
            const char* xrow = a.x + ((size_t)n * a.H + (rowok ? hi : 0)) * a.W * a.C * ES + c0 * ES;
            const char* drow = a.dy + ((size_t)p0 * a.K + n0 + lc * VEC) * ES;
            const int wbase = (int)wo0 * a.stride - a.pad_l + ts;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int prow = prow0 + RPS * i;
                int wi = wbase + prow * a.stride;
                bool ok = rowok && xcol_ok;
                if (a.reflect) wi = wi < 0 ? -wi : (wi >= a.W ? 2 * (a.W - 1) - wi : wi);
                else ok = ok && (unsigned)wi < (unsigned)a.W;
                const char* sx = ok ? xrow + (uint32_t)(wi * a.C * ES) : zero;
                const char* sd = dcol_ok ? drow + (uint32_t)(prow * a.K * ES) : zero;
                const int wrow = (wave * 64) / NCH + RPS * i;
                dma16_to_lds(sx, (__attribute__((address_space(3))) void*)(sX + wrow * PITCH));
                dma16_to_lds(sd, (__attribute__((address_space(3))) void*)(sD + wrow * PITCH));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int prow = prow0 + RPS * i, p = p0 + prow;
            const char* sx = zero;
            const char* sd = zero;
            if (p < pend) {
                if (xcol_ok) {
                    uint32_t n = fdiv((uint32_t)p, a.dHW), rem = (uint32_t)p - n * a.dHW.d;
                    uint32_t ho = fdiv(rem, a.dW), wo = rem - ho * a.dW.d;
                    int hi = (int)ho * a.stride - a.pad_t + tr, wi = (int)wo * a.stride - a.pad_l + ts;
                    bool ok = true;
                    if (a.reflect) {
                        hi = hi < 0 ? -hi : (hi >= a.H ? 2 * (a.H - 1) - hi : hi);
                        wi = wi < 0 ? -wi : (wi >= a.W ? 2 * (a.W - 1) - wi : wi);
                    } else ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                    if (ok) sx = a.x + ((((size_t)n * a.H + hi) * a.W + wi) * a.C + c0) * ES;
                }
                if (dcol_ok) sd = a.dy + ((size_t)p * a.K + n0 + lc * VEC) * ES;
            }
            // one wave-instruction = 64 lanes x 16 B = 1 KiB of consecutive tile bytes starting at the wave's first row
            const int wrow = (wave * 64) / NCH + RPS * i;
            dma16_to_lds(sx, (__attribute__((address_space(3))) void*)(sX + wrow * PITCH));
            dma16_to_lds(sd, (__attribute__((address_space(3))) void*)(sD + wrow * PITCH));
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, u = lane & 15;
    const int ksteps = pend > pbeg ? (pend - pbeg + BKP - 1) / BKP : 0;
    if (ksteps > 0) stage_tile(0, pbeg);
    SGG_WAIT_VM0();
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < ksteps) stage_tile(cur ^ 1, pbeg + (ks + 1) * BKP);
        const char* bX = smem + cur * STAGE;
        const char* bD = bX + TILE;
        if constexpr (sizeof(T) == 2) {
            // Fragment pipeline as in conv_gemm_glds_kernel: every dy fragment of the stage up front, x fragments in
            // groups of GJ fetched one group ahead of the MFMAs that consume them (the transposing reads otherwise sit
            // directly in front of their MFMAs and their latency is exposed whenever the DMA shares the LDS port).
            const int q = u >> 2, pp = u & 3;
            constexpr int KK = BKP / 32, GJ = 4, GPK = MI / GJ, NG = KK * GPK;
            auto ldT = [&](const char* base, int kk, int col) -> bf16x8 {
                const int r0 = kk * 32 + 8 * g + q, r1 = r0 + 4;
                const int k0 = wg2_key<T>(r0) & (NCH - 1), k1 = wg2_key<T>(r1) & (NCH - 1);
                const int ch = col >> 3, sub = (col & 7) * 2;
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(base + r0 * PITCH + ((ch ^ k0) << 4) + sub));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(base + r1 * PITCH + ((ch ^ k1) << 4) + sub));
                return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            bf16x8 fb[KK][NI];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[kk][j] = ldT(bD, kk, wn * 64 + j * 16 + 4 * pp);
            bf16x8 fa[2][GJ];
#pragma unroll
            for (int ii = 0; ii < GJ; ++ii) fa[0][ii] = ldT(bX, 0, wm * 128 + ii * 16 + 4 * pp);
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) {
                const int kk = gg / GPK, ib = (gg % GPK) * GJ;
                if (gg + 1 < NG) {
                    const int kn = (gg + 1) / GPK, in = ((gg + 1) % GPK) * GJ;
#pragma unroll
                    for (int ii = 0; ii < GJ; ++ii) fa[(gg + 1) & 1][ii] = ldT(bX, kn, wm * 128 + (in + ii) * 16 + 4 * pp);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ii = 0; ii < GJ; ++ii)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[ib + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[gg & 1][ii], acc[ib + ii][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s4 = 0; s4 < BKP / 4; ++s4) {
                const int row = 4 * s4 + g;
                const int key = wg2_key<T>(row) & (NCH - 1);
                float fb[NI];
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    int col = wn * 64 + j * 16 + u;
                    fb[j] = *reinterpret_cast<const float*>(bD + row * PITCH + (((col >> 2) ^ key) << 4) + (col & 3) * 4);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    int col = wm * 128 + i * 16 + u;
                    float fa = *reinterpret_cast<const float*>(bX + row * PITCH + (((col >> 2) ^ key) << 4) + (col & 3) * 4);
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j], fa, acc[i][j], 0, 0, 0);
                }
            }
        }
        SGG_WAIT_VM0();
        __syncthreads();
    }

    // The dy fragment is the MFMA A operand, so D[row = cout][col = (tap,c)]: lane (g,u) owns couts 4g..4g+3 of row u ->
    // one 16-byte store per fragment into the [row][cout] slab (4x fewer store instructions than the transposed layout)
    float* slab = a.ws + (size_t)split * Mrows * a.K;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int mr = m0 + wm * 128 + i * 16 + u;
        if (mr >= Mrows) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int k = n0 + wn * 64 + j * 16 + g * 4;
            if (k < a.K) *reinterpret_cast<f32x4*>(slab + (size_t)mr * a.K + k) = acc[i][j];
        }
    }
}

#ifndef SGG_NT_SLABS
#define SGG_NT_SLABS 1          // 1: the reducer reads the split slabs (their last use) with the streaming cache policy
#endif
#ifndef SGG_NT_ADDEND
#define SGG_NT_ADDEND 0         // 1: the 3x3 halo data gradient reads its skip-gradient addend (its last use) with the streaming policy
#endif
// dw[tap][c<Cr][k<Kr] (+)= sum_split ws[split][tap*C + c][k]   (fixed summation tree -> deterministic)
// One thread per 4 consecutive k (K is a multiple of 8): 16-byte slab reads, 4 independent partial sums in flight.
// dw2 != nullptr: two networks in one launch -- slabs [splits, 2 * splits) of ws sum into dw2
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* ws, float* dw, int taps, int C, int K, int Cr, int Kr, int splits, int accumulate,
                                                           float* dw2 = nullptr) {
    const int K4 = K / 4;
    const int64_t total1 = (int64_t)taps * Cr * K4, total = dw2 ? 2 * total1 : total1;
    const size_t slab = (size_t)taps * C * K;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += (int64_t)gridDim.x * blockDim.x) {
        const bool netb = i0 >= total1;
        const int64_t i = netb ? i0 - total1 : i0;
        if (netb) dw = dw2;                              // (i0 only grows: once in the second network, always there)
        int k4 = (int)(i % K4);
        int64_t t = i / K4;
        int c = (int)(t % Cr), tap = (int)(t / Cr);
        const float* src = ws + (netb ? (size_t)splits * slab : 0) + ((size_t)tap * C + c) * K + k4 * 4;
        // eight independent 16-byte loads in flight per thread (the grid is only ~9 waves per CU: with four the pass ran at 4.8 TB/s),
        // combined in a fixed tree -> deterministic
        f32x4 acc8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc8[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int sp = 0;
        for (; sp + 8 <= splits; sp += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = SGG_NT_SLABS ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)(sp + u) * slab))
                                    : *reinterpret_cast<const f32x4*>(src + (size_t)(sp + u) * slab);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc8[u] += v[u];
        }
        for (; sp < splits; ++sp) acc8[sp & 7] += *reinterpret_cast<const f32x4*>(src + (size_t)sp * slab);
        f32x4 s4 = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
        float* o = dw + ((size_t)tap * Cr + c) * Kr + k4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k4 * 4 + e < Kr) o[e] = accumulate ? o[e] + s4[e] : s4[e];
    }
}

// HWIO f32 -> [Kpad][R*S*Cpad] and [Cpad][R*S*Kpad] in dtype T, zero padded
template <typename T>
__global__ void pack_weights_kernel(const float* w, int taps, int C, int K, int Cpad, int Kpad, T* wf, T* wd) {
    int64_t total = (int64_t)taps * Cpad * Kpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int k = (int)(i % Kpad);
        int64_t t = i / Kpad;
        int c = (int)(t % Cpad), tap = (int)(t / Cpad);
        float v = (c < C && k < K) ? w[((size_t)tap * C + c) * K + k] : 0.f;
        if (wf) wf[((size_t)k * taps + tap) * Cpad + c] = (T)v;
        if (wd) wd[((size_t)c * taps + tap) * Kpad + k] = (T)v;
    }
}

// every conv layer of a network in ONE launch (blockIdx.y = layer): after each optimizer step all packed weights are stale,
// and 16 separate 8-us launches per network were pure launch latency
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const sgg_pack_item* items) {
    // (tap, 64 c, 64 k) tiles: w[tap][c][k] is read along k, w_dgrad[c][tap][k] written along k, and w_fwd[k][tap][c] -- the
    // transposed layout -- written along c out of an LDS tile.  (One thread per element wrote w_fwd as 2-byte scatters, one
    // cache line per lane: 83 us for a generator's 11.4 M parameters, 1.1 TB/s.)
    __shared__ float tile[64][65];
    const sgg_pack_item it = items[blockIdx.y];
    const int taps = it.taps, C = it.C, K = it.K, Cpad = it.Cpad, Kpad = it.Kpad;
    const float* w = it.w;
    T* wf = (T*)it.w_fwd;
    T* wd = (T*)it.w_dgrad;
    const int tc = (Cpad + 63) / 64, tk = (Kpad + 63) / 64;
    const int ntiles = taps * tc * tk;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int kt = t % tk, ct = (t / tk) % tc, tap = t / (tk * tc);
        const int k = kt * 64 + lx;
        for (int cy = ly; cy < 64; cy += 4) {
            const int c = ct * 64 + cy;
            const float v = (c < C && k < K) ? w[((size_t)tap * C + c) * K + k] : 0.f;
            tile[cy][lx] = v;
            if (wd && c < Cpad && k < Kpad) wd[((size_t)c * taps + tap) * Kpad + k] = (T)v;
        }
        __syncthreads();
        if (wf) {
            const int c = ct * 64 + lx;
            for (int ky = ly; ky < 64; ky += 4) {
                const int kk = kt * 64 + ky;
                if (c < Cpad && kk < Kpad) wf[((size_t)kk * taps + tap) * Cpad + c] = (T)tile[lx][ky];
            }
        }
        __syncthreads();
    }
}

static bool desc_ok(const sgg_conv_desc* d) {
    if (!d) return false;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->R <= 0 || d->S <= 0) return false;
    if (d->C <= 0 || d->K <= 0 || d->C % SGG_CPAD || d->K % SGG_CPAD) return false;
    if (d->stride != 1 && d->stride != 2) return false;
    if (d->dtype != SGG_F32 && d->dtype != SGG_BF16) return false;
    if (d->pad_mode == SGG_PAD_REFLECT) {
        if (d->stride != 1 || d->pad_t != d->pad_l || d->pad_t >= d->H || d->pad_t >= d->W) return false;
        if (d->Ho != d->H + 2 * d->pad_t - d->R + 1 || d->Wo != d->W + 2 * d->pad_l - d->S + 1) return false;
    } else if (d->pad_mode != SGG_PAD_ZERO) return false;
    if (d->pad_t < 0 || d->pad_l < 0) return false;
    // every output window must start inside the (leading-)padded input
    if ((int64_t)(d->Ho - 1) * d->stride - d->pad_t >= d->H || (int64_t)(d->Wo - 1) * d->stride - d->pad_l >= d->W) return false;
    if ((int64_t)d->N * d->H * d->W >= (1ll << 31) / 2 || (int64_t)d->N * d->Ho * d->Wo >= (1ll << 31) / 2) return false;
    return true;
}

static ConvArgs make_args(const sgg_conv_desc* d, const void* src, const void* w, const float* bias, void* dst, int act, float leak) {
    ConvArgs a;
    a.src = (const char*)src; a.wmat = (const char*)w; a.bias = bias; a.dst = (char*)dst; a.addend = nullptr; a.fold = nullptr; a.stats = nullptr; a.nx = nullptr; a.nstats = nullptr; a.ngamma = nullptr; a.nbeta = nullptr; a.nact = 0; a.nleak = 0.f; a.nout = nullptr; a.ngamma2 = nullptr; a.nbeta2 = nullptr; a.partial = nullptr; a.ksplit = 1; a.pdst = 0; a.dst_f32 = 0; a.addend_f32 = 0; a.wmat2 = nullptr; a.bias2 = nullptr; a.nsplit = 0x7fffffff;
    a.grp = 1; a.net_src = a.net_dst = a.net_add = a.net_part = 0;
    a.ablate = sgg_config().ablate;
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.K = d->K; a.R = d->R; a.S = d->S; a.stride = d->stride;
    a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.Ho = d->Ho; a.Wo = d->Wo; a.reflect = d->pad_mode == SGG_PAD_REFLECT;
    a.act = act; a.leak = leak;
    return a;
}

template <typename T, int MODE, int BM, int BN, int WGM>
static int launch_gemm_cfg(const ConvArgs& a, int64_t Mmax, int DC, int classes, hipStream_t s) {
    constexpr size_t lds = 2 * (BM + BN) * 128;
    auto kern = conv_gemm_kernel<T, MODE, BM, BN, WGM>;
    SGG_LDS_ATTR(kern, (int)lds);
    dim3 grid((unsigned)((Mmax + BM - 1) / BM), (unsigned)((DC + BN - 1) / BN), (unsigned)classes);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return sgg_check_launch();
}

// out[p][c] = act(sum_s partial[s][p][c] + bias[c])  (fixed order -> deterministic)
struct SplitKGroup { const float* bias2; size_t net_part, net_dst, net_add; };   // second group of a grouped launch (blockIdx.y == 1)
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* partial, const float* bias, const char* addend, char* out, int64_t nvec4,
                                                            int DC, int ksplit, size_t slab, int act, float leak, SplitKGroup g2) {
    if (blockIdx.y) {
        partial = reinterpret_cast<const float*>(reinterpret_cast<const char*>(partial) + g2.net_part);
        bias = g2.bias2; out += g2.net_dst;
        if (addend) addend += g2.net_add;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 s4 = *reinterpret_cast<const f32x4*>(partial + i * 4);
        for (int sp = 1; sp < ksplit; ++sp) s4 += *reinterpret_cast<const f32x4*>(partial + sp * slab + i * 4);
        int c = (int)((i * 4) % DC);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = s4[e] + (bias ? bias[c + e] : 0.f);
        act_dispatch(act, [&](auto act_c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_apply_c<decltype(act_c)::value>(v[e], leak);
        });
        if (addend) {
            const T* ad = reinterpret_cast<const T*>(addend) + i * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)ad[e];
        }
        if constexpr (sizeof(T) == 2) {
            bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *reinterpret_cast<bf16x4*>(out + i * 8) = pk;
        } else *reinterpret_cast<f32x4*>(out + i * 16) = (f32x4){v[0], v[1], v[2], v[3]};
    }
}

// split-K factor for layers whose output is too small to fill the chip (D's tail: 5x13..15x31 maps, K = 4608)
static int conv_ksplit(int64_t Mmax, int DC, int classes, int ktot_max) {
    int64_t bm = 128, bn = DC >= 128 ? 128 : (DC > 16 ? 64 : 16);
    int64_t blocks = ((Mmax + bm - 1) / bm) * ((DC + bn - 1) / bn) * classes;
    if (blocks >= 192 || ktot_max < 16) return 1;
    int64_t ks = (256 + blocks - 1) / blocks;
    if (ks > ktot_max / 4) ks = ktot_max / 4;
    if (ks > 16) ks = 16;
    return ks < 2 ? 1 : (int)ks;
}

template <typename T, int MODE, int BM, int BN, int WGM, int NW, int BKB = 128, int NS = 2>
static int launch_glds_cfg(const ConvArgs& a, int64_t Mmax, int DC, int classes, hipStream_t s) {
    constexpr int RPP = NW * (64 / (BKB / 16));
    constexpr size_t lds = (size_t)NS * (size_t)(BM + (BN + RPP - 1) / RPP * RPP) * BKB;
    auto kern = conv_gemm_glds_kernel<T, MODE, BM, BN, WGM, NW, BKB, NS>;
    SGG_LDS_ATTR(kern, (int)lds);
    const int64_t tilesN = (DC + BN - 1) / BN;
    int64_t blocks = (Mmax + BM - 1) / BM * tilesN;
    dim3 grid((unsigned)blocks, (unsigned)a.ksplit, (unsigned)(classes * a.grp));
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
    return sgg_check_launch();
}

// SGG_CONV_IMPL=reg selects the v1 register-staged kernels (kept for A/B runs); default = v2 direct-to-LDS
static bool use_glds() { return sgg_config().glds != 0; }

#ifndef GLDS_NS_256x128
#define GLDS_NS_256x128 3                              // LDS stages of the 256x128 tile (48 KB each): its layers (c2 forward, d2 data gradient,
                                                       // D.h1 forward) gather their pixel operand from HBM with 9 short K-tiles per block;
                                                       // two tiles in flight instead of one: 96.8 -> 93.4 us (c2), 26.4 -> 25.3 us (D.h1)
#endif
template <typename T, int MODE>
static int launch_gemm(const ConvArgs& a, hipStream_t s) {
    int DC, classes;
    int64_t Mmax;
    if (MODE == MODE_FWD) { DC = a.K; classes = 1; Mmax = (int64_t)a.N * a.Ho * a.Wo; }
    else if (MODE == MODE_BORDER) {
        DC = a.C; classes = 1;
        int p2 = 2 * a.pad_t;
        bool all = a.H <= p2 + 1 || a.W <= p2 + 1;
        Mmax = (int64_t)a.N * (all ? a.H * a.W : p2 * a.W + (a.H - p2) * p2);
    } else {
        DC = a.C; classes = a.stride * a.stride;
        Mmax = (int64_t)a.N * ((a.H + a.stride - 1) / a.stride) * ((a.W + a.stride - 1) / a.stride);
    }
    if constexpr (MODE != MODE_BORDER && sizeof(T) == 2) {
        // grouped launch of a shape whose kernel takes a STACKED batch with per-image weights (the LDS-resident 3x3 / stride-2
        // halo kernels): the two groups are contiguous, so it is that kernel on 2N images with the split at N
        if (a.grp == 2 && a.ksplit == 1 && (halo3_ok(a, MODE, true) || (MODE == MODE_DGRAD && s2halo_ok(a, true)))) {
            ConvArgs b = a;
            b.N = 2 * a.N; b.nsplit = a.N; b.grp = 1;
            return launch_gemm<T, MODE>(b, s);
        }
    }
    if constexpr (MODE == MODE_DGRAD && sizeof(T) == 2) {
        if (s2halo_ok(a, true)) return launch_s2halo(a, s);
    }
    if constexpr (MODE != MODE_BORDER && sizeof(T) == 2) {
        if (halo3_ok(a, MODE, true)) {
            if constexpr (MODE == MODE_DGRAD) {
                const bool pair = a.wmat2 != nullptr;
                if (a.stats && (pair || a.dst_f32 || a.addend_f32)) return SGG_EUNSUPPORTED;   // (the norm-backward sums have no paired / mixed form)
                if (a.reflect) {
                    if (a.stats) return launch_halo3<MODE_DGRAD, true, 2>(a, s);
                    return pair ? launch_halo3<MODE_DGRAD, true, 0, true>(a, s) : launch_halo3<MODE_DGRAD, true>(a, s);
                }
                if (a.stats) return launch_halo3<MODE_DGRAD, false, 2>(a, s);
                if (pair) return launch_halo3<MODE_DGRAD, false, 0, true>(a, s);
            }
            if constexpr (MODE == MODE_FWD) {
                if (a.nout) {
                    if (!a.stats || !a.reflect || a.C > H3_NORM_MAXC) return SGG_EUNSUPPORTED;
                    return a.wmat2 ? launch_halo3<MODE_FWD, false, 1, true, 2>(a, s) : launch_halo3<MODE_FWD, false, 1, false, 2>(a, s);
                }
                if (a.reflect) {
                    if (a.wmat2) return a.stats ? launch_halo3<MODE_FWD, false, 1, true, 1>(a, s) : launch_halo3<MODE_FWD, false, 0, true, 1>(a, s);
                    return a.stats ? launch_halo3<MODE_FWD, false, 1, false, 1>(a, s) : launch_halo3<MODE_FWD, false, 0, false, 1>(a, s);
                }
                if (a.wmat2) return a.stats ? launch_halo3<MODE_FWD, false, 1, true>(a, s) : launch_halo3<MODE_FWD, false, 0, true>(a, s);
                if (a.stats) return launch_halo3<MODE_FWD, false, 1>(a, s);
            }
            return launch_halo3<MODE, false>(a, s);
        }
    }
    if constexpr (MODE != MODE_BORDER) {
        if (use_glds()) {
            // big square tiles (8 waves, 1 block per CU) when there is enough work to fill the chip with them
            // (a 4-stage ring of 64-byte K-slices was measured 3-5 % slower on the residual conv and is not built)
            if (DC >= 256 && Mmax * ((DC + 255) / 256) >= 256 * 160)
                return launch_glds_cfg<T, MODE, 256, 256, 2, 8, 128, 2>(a, Mmax, DC, classes, s);
            // short reductions (<= 16 K-tiles: the 64-channel stride-2 layers c2 forward / d2 data gradient / D.h1 forward): a
            // 256x128 block is alone on its CU and spends a third of its life in prologue and epilogue; two 128x128 blocks per
            // CU cover each other: 93.7 -> 84.4 us (c2), 25.3 -> 22.7 us (D.h1); the 4-wave 128x64 tile (three blocks): 115 us
            if (DC >= 128 && (MODE == MODE_FWD ? a.R * a.S * a.C : a.R * a.S * a.K) <= 1024)
                return launch_glds_cfg<T, MODE, 128, 128, 2, 8>(a, Mmax, DC, classes, s);
            if (DC >= 128 && Mmax * ((DC + 127) / 128) >= 256 * 160)
                return launch_glds_cfg<T, MODE, 256, 128, 4, 8, 128, GLDS_NS_256x128>(a, Mmax, DC, classes, s);
            // 128x128 with 8 waves (2 per SIMD, two blocks per CU): the 4-wave variant ran at one wave per SIMD with
            // nothing to cover its LDS latencies (D.h3 data gradient 86 -> 60 us, D.h2 forward 36 -> 23 us)
            if (DC >= 128) return launch_glds_cfg<T, MODE, 128, 128, 2, 8>(a, Mmax, DC, classes, s);
            if (DC > 16) return launch_glds_cfg<T, MODE, 128, 64, 4, 4>(a, Mmax, DC, classes, s);
            return launch_glds_cfg<T, MODE, 256, 16, 4, 4>(a, Mmax, DC, classes, s);
        }
    }
    if constexpr (MODE == MODE_BORDER) {               // few pixels, short K loops: small tiles for parallelism
        if (DC >= 128) return launch_gemm_cfg<T, MODE, 64, 128, 2>(a, Mmax, DC, classes, s);
        if (DC > 16) return launch_gemm_cfg<T, MODE, 64, 64, 2>(a, Mmax, DC, classes, s);
        return launch_gemm_cfg<T, MODE, 64, 16, 4>(a, Mmax, DC, classes, s);
    }
    if (DC >= 128) return launch_gemm_cfg<T, MODE, 128, 128, 2>(a, Mmax, DC, classes, s);
    if (DC > 16) return launch_gemm_cfg<T, MODE, 128, 64, 4>(a, Mmax, DC, classes, s);
    return launch_gemm_cfg<T, MODE, 256, 16, 4>(a, Mmax, DC, classes, s);
}

// split-K planning shared by the workspace queries and the launches
struct GemmPlan { int DC, classes, ktot_max, ksplit; int64_t Mmax; size_t pdst, ws_bytes; };
static GemmPlan plan_gemm(const sgg_conv_desc* d, int mode) {
    GemmPlan g;
    const int vec = d->dtype == SGG_BF16 ? 8 : 4;
    if (mode == MODE_FWD) {
        g.DC = d->K; g.classes = 1; g.Mmax = (int64_t)d->N * d->Ho * d->Wo; g.pdst = (size_t)g.Mmax;
        g.ktot_max = (d->R * d->S * (d->C / vec) + 7) / 8;
    } else {
        const int st = d->stride;
        g.DC = d->C; g.classes = st * st;
        g.Mmax = (int64_t)d->N * ((d->H + st - 1) / st) * ((d->W + st - 1) / st);
        g.pdst = (size_t)d->N * d->H * d->W;
        g.ktot_max = (((d->R + st - 1) / st) * ((d->S + st - 1) / st) * (d->K / vec) + 7) / 8;
    }
    g.ksplit = use_glds() ? conv_ksplit(g.Mmax, g.DC, g.classes, g.ktot_max) : 1;
    if (g.ksplit > 1 && (mode == MODE_FWD || mode == MODE_DGRAD)) {
        // shapes the halo-resident kernels take are never split (they only occur at sizes that fill the chip; it also
        // keeps the small parity-test shapes on the same kernels as the full-size layers)
        ConvArgs a = make_args(d, nullptr, nullptr, nullptr, nullptr, SGG_ACT_NONE, 0.f);
        const bool bf = d->dtype == SGG_BF16;
        if (halo3_ok(a, mode, bf) || (mode == MODE_DGRAD && s2halo_ok(a, bf))) g.ksplit = 1;
    }
    g.ws_bytes = g.ksplit > 1 ? (size_t)g.ksplit * g.pdst * g.DC * sizeof(float) : 0;
    return g;
}

template <typename T, int MODE>
static int run_gemm(const sgg_conv_desc* d, ConvArgs a, void* ws, size_t ws_bytes, hipStream_t s) {
    GemmPlan g = plan_gemm(d, MODE);
    if (g.ksplit > 1) {
        if (!ws || ws_bytes < g.ws_bytes * a.grp) return SGG_EWORKSPACE;
        a.ksplit = g.ksplit; a.partial = (float*)ws; a.pdst = g.pdst; a.net_part = g.ws_bytes;
    }
    int rc = launch_gemm<T, MODE>(a, s);
    if (rc || g.ksplit <= 1) return rc;
    int64_t nvec4 = (int64_t)g.pdst * g.DC / 4;
    int blocks = (int)((nvec4 + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(blocks, a.grp), dim3(256), 0, s, (const float*)ws, a.bias, a.addend, a.dst, nvec4, g.DC, g.ksplit,
                       g.pdst * g.DC, a.act, a.leak, SplitKGroup{a.bias2, a.net_part, a.net_dst, a.net_add});
    return sgg_check_launch();
}

// -------------------------------------------------------------------------------------------------
// 3x3 stride-1 weight gradient with the x HALO resident in LDS and ALL 9 TAPS accumulated by one block (the residual
// blocks' convs, module.py:210-216 backward).
//
// conv_wgrad_glds_kernel gives every tap its own block, so a stage streams 32 KB of x and 32 KB of dy per 8.4 MFLOP
// and the kernel is bound by that L2->LDS stream, not by the MFMAs.  Here a block owns
// dW[9 taps][64 c][128 k] (36 MFMA tiles per wave: 9 taps x 4 cout tiles of its 16-channel slice) and walks pixel
// tiles of 2 rows x 64 columns: per stage the 4 x 66 halo pixels of its 64 input channels (36 KB) serve all nine taps
// as shifted rows, next to a [128 px][128 k] dy tile (32 KB) -- 68 KB per 18.9 MFLOP, 2.1x the FLOP per staged byte.
// Operands are transposed on the fly with ds_read_b64_tr_b16 as in conv_wgrad_glds_kernel; dy keeps that kernel's
// swizzle (256-byte rows), the 128-byte halo rows use w9_xkey (below), conflict-free for any tap shift.
// Pixel tiles are split over `splits` blocks per output tile; slabs are summed in fixed order by wgrad_reduce_kernel.
// -------------------------------------------------------------------------------------------------
#define W9_TW 64
#define W9_PITCH 72                                    // halo row pitch in pixels (66 used; multiple of 8 = one DMA)
#define W9_XBYTES (4 * W9_PITCH * 128)
#define W9_DBYTES (128 * 256)
#define W9_STAGE (W9_XBYTES + W9_DBYTES)

struct W9Args {
    const char* x;       // (N,H,W,C)
    const char* dy;      // (N,H,W,K)
    const char* x2;      // optional second (x, dy) pair of the same shape: pixel tiles [tiles/2, tiles) read it, so the two
    const char* dy2;     //   applications of one layer in a step share one launch, one set of slabs and one reduce
    float* ws;           // [splits][9*C][K] f32 slabs
    int N, H, W, C, K, reflect;
    int tiles, tiles_per_split;   // tiles counts both pairs
    // Two NETWORKS of one shape in one launch (the cycle step's G_A->B beside G_B->A): splits [splits_per_net, 2 * splits_per_net)
    // belong to the second network and read xb / dyb / x2b / dy2b; each network then gets half the blocks, i.e. half the slabs
    // to write and to reduce for the same work per CU.  splits_per_net >= the grid's split count: one network.
    const char* xb; const char* dyb; const char* x2b; const char* dy2b;
    int splits_per_net;
};

// 16-byte chunk XOR key of halo row `row` (128-byte rows, two per 256-byte bank line): a transposing read touches rows
// {b..b+3, b+8..b+11} per 32-lane group; bits 0, 1 and 3 of the row index tell those eight apart for every b, bit 0
// already selects the half line, bits 1 and 3 pick one of the four 32-byte column pairs.
__device__ inline int w9_xkey(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }

#ifndef W9_NT
#define W9_NT 0          // streaming (nt) LDS-DMA of the operands: 1 the x halo (saved forward activations: their last use), 2 the dy tiles
#endif
__global__ __launch_bounds__(512) void conv3x3_wgrad_halo_kernel(W9Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages of { X halo [4][72][128 B], DY [128][256 B] }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index as a scalar
    const int ct = wave & 3, kh = wave >> 2;             // wave: 16-channel slice of the 64, 64-cout half of the 128
    const int ktiles = a.K >> 7, otiles = (a.C >> 6) * ktiles;
    int lid;
    {
        const int b = (int)blockIdx.x, nm = (int)gridDim.x;
        const int q = nm >> 3, rr = nm & 7, xcd = b & 7;
        lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
    }
    const int split = lid / otiles, tl = lid - split * otiles;
    const int c0 = (tl / ktiles) * 64, n0 = (tl % ktiles) * 128;
    const bool netb = split >= a.splits_per_net;         // second network's blocks (uniform per block)
    const int nsplit = netb ? split - a.splits_per_net : split;
    const int t_beg = nsplit * a.tiles_per_split;
    const int t_end = min(a.tiles, t_beg + a.tiles_per_split);
    const int tilesW = a.W / W9_TW, tilesH = a.H >> 1;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    const char* const X1 = netb ? a.xb : a.x;
    const char* const D1 = netb ? a.dyb : a.dy;
    const char* const X2 = netb ? a.x2b : a.x2;
    const char* const D2 = netb ? a.dy2b : a.dy2;

    const int tiles1 = X2 ? a.tiles >> 1 : a.tiles;      // tiles of the first (x, dy) pair
    auto stage_tile = [&](int stg, int tt) {
        const bool second = tt >= tiles1;
        const int t = second ? tt - tiles1 : tt;
        const char* xs = second ? X2 : X1;
        const char* dys = second ? D2 : D1;
        const int tw = t % tilesW, rest = t / tilesW;
        const int th = rest % tilesH, n = rest / tilesH;
        const int h0 = th * 2, w0 = tw * W9_TW;
        char* sX = smem + stg * W9_STAGE;
        char* sD = sX + W9_XBYTES;
        // dy: 32 wave-instructions of 4 pixels x 256 B; wave w issues 4w..4w+3
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = wave * 4 + i;
            const int px = d * 4 + (lane >> 4), pos = lane & 15;
            const int key = wg2_key<bf16>(px) & 15;
            const int trow = px >> 6, tcol = px & 63;
            const char* src = dys + ((((size_t)n * a.H + h0 + trow) * a.W + w0 + tcol) * a.K + n0) * 2 + ((pos ^ key) << 4);
            if (W9_NT & 2) dma16_to_lds_nt(src, (__attribute__((address_space(3))) void*)(sD + d * 1024));
            else dma16_to_lds(src, (__attribute__((address_space(3))) void*)(sD + d * 1024));
        }
        // x halo: 36 wave-instructions of 8 pixels x 128 B (row k = q / 9, column group q % 9); wave w issues w, w+8, ...
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = wave + 8 * i;
            if (q >= 36) break;
            const int k = q / 9, cg = q - 9 * k;
            const int hp = cg * 8 + (lane >> 3), pos = lane & 7;
            const int key = w9_xkey(k * W9_PITCH + hp);
            int hi = h0 - 1 + k, wi = w0 - 1 + hp;
            bool ok = hp < W9_TW + 2;
            if (a.reflect) {
                hi = hi < 0 ? -hi : (hi >= a.H ? 2 * (a.H - 1) - hi : hi);
                wi = wi < 0 ? -wi : (wi >= a.W ? 2 * (a.W - 1) - wi : wi);
            } else ok = ok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const char* src = ok ? xs + ((((size_t)n * a.H + hi) * a.W + wi) * a.C + c0) * 2 + ((pos ^ key) << 4) : zero;
            if (W9_NT & 1) dma16_to_lds_nt(src, (__attribute__((address_space(3))) void*)(sX + (k * W9_PITCH + cg * 8) * 128));
            else dma16_to_lds(src, (__attribute__((address_space(3))) void*)(sX + (k * W9_PITCH + cg * 8) * 128));
        }
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, u = lane & 15, q4 = u >> 2, pp = u & 3;
    // Fragment addresses = stage base + compile-time constant + lane part.  The swizzle keys depend on the low 4 bits of
    // the row index only: for dy that is the lane's row within the 32-pixel step; for the halo it is (b + lane row) mod 16
    // with b mod 16 = 8 * (halo row parity) + tap column, i.e. six classes.
    int doff[4][2], xoff[6][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int L = 8 * g + q4 + 4 * h;
        const int kd = wg2_key<bf16>(L) & 15;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int col = kh * 64 + kt * 16 + 4 * pp;
            doff[kt][h] = L * 256 + (((col >> 3) ^ kd) << 4) + (col & 7) * 2;
        }
#pragma unroll
        for (int cls = 0; cls < 6; ++cls) {
            const int bl = (cls / 3) * 8 + (cls % 3);
            const int col = ct * 16 + 4 * pp;
            xoff[cls][h] = L * 128 + (((col >> 3) ^ w9_xkey(bl + L)) << 4) + (col & 7) * 2;
        }
    }
    if (t_beg < t_end) stage_tile(0, t_beg);
    SGG_WAIT_VM0();
    __builtin_amdgcn_s_barrier();
    for (int t = t_beg; t < t_end; ++t) {
        const int cur = (t - t_beg) & 1;
        if (t + 1 < t_end) stage_tile(cur ^ 1, t + 1);
        const char* bX = smem + cur * W9_STAGE;
        const char* bD = bX + W9_XBYTES;
        // lane-varying address parts are precomputed (xoff / doff); what changes per fragment is a compile-time constant
        auto ldD = [&](int kk, int kt) -> bf16x8 {      // dy fragment: 32 pixels of k32-step kk x 16 couts
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(bD + kk * (32 * 256) + doff[kt][0]));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(bD + kk * (32 * 256) + doff[kt][1]));
            return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto ldX = [&](int kk, int tap) -> bf16x8 {     // x fragment of tap (r, s): the same 32 pixels shifted in the halo
            const int r = tap / 3, sx = tap - 3 * r;
            const int b = ((kk >> 1) + r) * W9_PITCH + (kk & 1) * 32 + sx;       // first halo row of the fragment
            const int cls = ((((kk >> 1) + r) & 1) * 3) + sx;                     // b mod 16 = 8 * parity + sx
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(bX + b * 128 + xoff[cls][0]));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(bX + b * 128 + xoff[cls][1]));
            return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        // 36 steps (k32-step kk, tap) of 4 MFMAs; x fragments rotate through 3 registers sets and are fetched two steps
        // (128 MFMA cycles) ahead, the dy fragments of the next k32-step during the last taps of the current one
        bf16x8 fd[2][4], fx[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) fd[0][j] = ldD(0, j);
        fx[0] = ldX(0, 0);
        fx[1] = ldX(0, 1);
#pragma unroll
        for (int i = 0; i < 36; ++i) {
            const int kk = i / 9, tap = i - 9 * kk;
            if (i + 2 < 36) fx[(i + 2) % 3] = ldX((i + 2) / 9, (i + 2) % 9);
            if (tap >= 4 && tap < 8 && kk < 3) fd[(kk + 1) & 1][tap - 4] = ldD(kk + 1, tap - 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[tap][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[kk & 1][j], fx[i % 3], acc[tap][j], 0, 0, 0);
        }
        SGG_WAIT_VM0();
        SGG_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
    }

    // D[row = cout][col = c]: lane (g,u) owns couts 4g..4g+3 of channel u -> one 16-byte store per tile
    float* slab = a.ws + (size_t)split * 9 * a.C * a.K;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const size_t mr = (size_t)tap * a.C + c0 + ct * 16 + u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = n0 + kh * 64 + j * 16 + g * 4;
            *reinterpret_cast<f32x4*>(slab + mr * a.K + k) = acc[tap][j];
        }
    }
}

// -------------------------------------------------------------------------------------------------
// The same for STRIDE 2 (c2 / c3 and, with the operand roles swapped, the Conv2DTranspose layers d1 / d2,
// module.py:236-242,254-260 backward).  One tap per block made these layers stream x nine times through L2
// (1.5 GB per launch at the c2 shape, 11.6 TB/s: L2-bound at 131 us).  Here a stage is one dy row segment of 64 pixels:
// the 3 x 129 input pixels it touches are staged once, de-interleaved by column parity (LDS slot = parity * 68 + column / 2),
// so that a tap's stride-2 walk over columns is a unit-stride walk over LDS rows and the transposing reads of
// conv3x3_wgrad_halo_kernel (and its swizzle) apply unchanged: tap column s reads plane s & 1 from row s >> 1.
// -------------------------------------------------------------------------------------------------
#define W9S_PITCH 136                                  // slots per halo row: 2 parity planes x 68 (65 / 64 used)
#define W9S_XBYTES (3 * W9S_PITCH * 128)               // 52224
#define W9S_DBYTES (64 * 256)                          // 16384
#define W9S_STAGE (W9S_XBYTES + W9S_DBYTES)

struct W9SArgs {
    const char* x;       // (N,H,W,C)
    const char* dy;      // (N,Ho,Wo,K)
    float* ws;           // [splits][9*C][K] f32 slabs
    int N, H, W, C, K, Ho, Wo, pad_t, pad_l;
    int tiles, tiles_per_split;
    // two NETWORKS of one shape in one launch (as W9Args): splits [splits_per_net, 2 * splits_per_net) read xb / dyb
    const char* xb; const char* dyb;
    int splits_per_net;
};

__global__ __launch_bounds__(512) void conv3x3_wgrad_halo_s2_kernel(W9SArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages of { X halo [3][136][128 B], DY [64][256 B] }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave & 3, kh = wave >> 2;
    const int ktiles = a.K >> 7, otiles = (a.C >> 6) * ktiles;
    int lid;
    {
        const int b = (int)blockIdx.x, nm = (int)gridDim.x;
        const int q = nm >> 3, rr = nm & 7, xcd = b & 7;
        lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
    }
    const int split = lid / otiles, tl = lid - split * otiles;
    const int c0 = (tl / ktiles) * 64, n0 = (tl % ktiles) * 128;
    const bool netb = split >= a.splits_per_net;          // second network's blocks (uniform per block)
    const int nsplit = netb ? split - a.splits_per_net : split;
    const char* const ax = netb ? a.xb : a.x;
    const char* const ady = netb ? a.dyb : a.dy;
    const int t_beg = nsplit * a.tiles_per_split;
    const int t_end = min(a.tiles, t_beg + a.tiles_per_split);
    const int tilesW = a.Wo / 64;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    auto stage_tile = [&](int stg, int t) {
        const int tw = t % tilesW, rest = t / tilesW;
        const int ho = rest % a.Ho, n = rest / a.Ho;
        const int w0 = tw * 64;
        char* sX = smem + stg * W9S_STAGE;
        char* sD = sX + W9S_XBYTES;
        // dy: 16 wave-instructions of 4 pixels x 256 B; wave w issues 2w, 2w+1
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int d = wave * 2 + i;
            const int px = d * 4 + (lane >> 4), pos = lane & 15;
            const int key = wg2_key<bf16>(px) & 15;
            const char* src = ady + ((((size_t)n * a.Ho + ho) * a.Wo + w0 + px) * a.K + n0) * 2 + ((pos ^ key) << 4);
            dma16_to_lds(src, (__attribute__((address_space(3))) void*)(sD + d * 1024));
        }
        // x halo: 3 rows x 17 wave-instructions of 8 slots; wave w issues q = w, w+8, ...
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int q = wave + 8 * i;
            if (q >= 51) break;
            const int k = q / 17, sg = q - 17 * k;
            const int slot = sg * 8 + (lane >> 3), pos = lane & 7;      // 0..135
            const int plane = slot >= 68 ? 1 : 0, j = slot - 68 * plane;
            const int col = 2 * j + plane;                              // halo column 0..128
            const int key = w9_xkey(k * W9S_PITCH + slot);
            const int hi = 2 * ho - a.pad_t + k, wi = 2 * w0 - a.pad_l + col;
            const bool ok = col <= 128 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const char* src = ok ? ax + ((((size_t)n * a.H + hi) * a.W + wi) * a.C + c0) * 2 + ((pos ^ key) << 4) : zero;
            dma16_to_lds(src, (__attribute__((address_space(3))) void*)(sX + (k * W9S_PITCH + sg * 8) * 128));
        }
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, u = lane & 15, q4 = u >> 2, pp = u & 3;
    // lane parts of the fragment addresses; halo classes by (b mod 16) = 8 * (r & 1) + 4 * (s & 1) + (s >> 1)
    int doff[4][2], xoff[6][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int L = 8 * g + q4 + 4 * h;
        const int kd = wg2_key<bf16>(L) & 15;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int col = kh * 64 + kt * 16 + 4 * pp;
            doff[kt][h] = L * 256 + (((col >> 3) ^ kd) << 4) + (col & 7) * 2;
        }
#pragma unroll
        for (int cls = 0; cls < 6; ++cls) {
            const int sx = cls % 3;
            const int bl = (cls / 3) * 8 + (sx & 1) * 4 + (sx >> 1);
            const int col = ct * 16 + 4 * pp;
            xoff[cls][h] = L * 128 + (((col >> 3) ^ w9_xkey(bl + L)) << 4) + (col & 7) * 2;
        }
    }
    if (t_beg < t_end) stage_tile(0, t_beg);
    SGG_WAIT_VM0();
    __builtin_amdgcn_s_barrier();
    for (int t = t_beg; t < t_end; ++t) {
        const int cur = (t - t_beg) & 1;
        if (t + 1 < t_end) stage_tile(cur ^ 1, t + 1);
        const char* bX = smem + cur * W9S_STAGE;
        const char* bD = bX + W9S_XBYTES;
        auto ldD = [&](int kk, int kt) -> bf16x8 {
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(bD + kk * (32 * 256) + doff[kt][0]));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(bD + kk * (32 * 256) + doff[kt][1]));
            return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto ldX = [&](int kk, int tap) -> bf16x8 {     // 32 dy pixels kk*32.. of the row <-> input columns 2*px + s, row r
            const int r = tap / 3, sx = tap - 3 * r;
            const int b = r * W9S_PITCH + (sx & 1) * 68 + (sx >> 1) + kk * 32;   // first LDS row of the fragment
            const int cls = (r & 1) * 3 + sx;
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(bX + b * 128 + xoff[cls][0]));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(bX + b * 128 + xoff[cls][1]));
            return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        // 18 steps (k32-step kk, tap) of 4 MFMAs; x fragments rotate through 3 register sets, fetched two steps ahead
        bf16x8 fd[2][4], fx[3];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 4; ++j) fd[kk][j] = ldD(kk, j);
        fx[0] = ldX(0, 0);
        fx[1] = ldX(0, 1);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int kk = i / 9, tap = i - 9 * kk;
            if (i + 2 < 18) fx[(i + 2) % 3] = ldX((i + 2) / 9, (i + 2) % 9);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[tap][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[kk][j], fx[i % 3], acc[tap][j], 0, 0, 0);
        }
        SGG_WAIT_VM0();
        SGG_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
    }

    float* slab = a.ws + (size_t)split * 9 * a.C * a.K;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const size_t mr = (size_t)tap * a.C + c0 + ct * 16 + u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = n0 + kh * 64 + j * 16 + g * 4;
            *reinterpret_cast<f32x4*>(slab + mr * a.K + k) = acc[tap][j];
        }
    }
}

static bool w9s_ok(const sgg_conv_desc* d) {
    const int en = sgg_config().w9s2;
    if (!en || !use_glds() || d->dtype != SGG_BF16 || d->pad_mode != SGG_PAD_ZERO) return false;
    if (d->R != 3 || d->S != 3 || d->stride != 2 || d->H != 2 * d->Ho || d->W != 2 * d->Wo) return false;
    if ((unsigned)d->pad_t > 1u || (unsigned)d->pad_l > 1u) return false;
    // K >= 256 layers (c3 / d1) already run the 256x256 DMA kernel at the same speed (72 us); this kernel is for the
    // 64 <-> 128 channel layers at full resolution, which were L2-bound at 131 us (89 us here)
    if (d->K >= 256 && d->R * d->S * d->C >= 256) return false;
    return d->Wo % 64 == 0 && d->C % 64 == 0 && d->K % 128 == 0;
}
static int w9s_tiles(const sgg_conv_desc* d) { return d->N * d->Ho * (d->Wo / 64); }
static int w9s_splits(const sgg_conv_desc* d) {
    const int otiles = (d->C / 64) * (d->K / 128), T = w9s_tiles(d);
    int sp = otiles >= 256 ? 1 : 256 / otiles;
    if (sp > T) sp = T;
    if (sp > 256) sp = 256;
    const int tps = (T + sp - 1) / sp;
    return (T + tps - 1) / tps;
}

static bool w9_ok(const sgg_conv_desc* d) {
    const int en = sgg_config().w9;
    if (!en || !use_glds() || d->dtype != SGG_BF16) return false;
    if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1 || d->Ho != d->H || d->Wo != d->W) return false;
    return d->W % W9_TW == 0 && d->H % 2 == 0 && d->C % 64 == 0 && d->K % 128 == 0;
}
static int w9_tiles(const sgg_conv_desc* d) { return d->N * (d->H / 2) * (d->W / W9_TW); }
// the split count depends on the single-pair tile count only, so a paired launch needs the same workspace
static int w9_splits(const sgg_conv_desc* d) {
    const int otiles = (d->C / 64) * (d->K / 128), T = w9_tiles(d);
    int sp = otiles >= 256 ? 1 : 256 / otiles;
    if (sp > T) sp = T;
    if (sp > 64) sp = 64;
    const int tps = (T + sp - 1) / sp;
    return (T + tps - 1) / tps;
}

template <typename T, int BMW, int BNW, int WGM>
static int launch_wgrad_cfg(WgradArgs& a, int splits, hipStream_t s) {
    constexpr size_t lds = 2 * 32 * (BMW + BNW) * sizeof(T);
    auto kern = conv_wgrad_kernel<T, BMW, BNW, WGM>;
    SGG_LDS_ATTR(kern, (int)lds);
    int Mrows = a.R * a.S * a.C;
    dim3 grid((unsigned)((Mrows + BMW - 1) / BMW), (unsigned)((a.K + BNW - 1) / BNW), (unsigned)splits);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return sgg_check_launch();
}

static bool wgrad_use_v2(const sgg_conv_desc* d) {
    return use_glds() && d->K >= 256 && d->R * d->S * d->C >= 256;
}

static int wgrad_splits(const sgg_conv_desc* d) {
    int64_t P = (int64_t)d->N * d->Ho * d->Wo;
    if (halo_wgrad_ok(d)) return halo_wgrad_blocks(d);               // one slab per persistent block
    if (s2n_wgrad_ok(d)) return s2n_wgrad_blocks(d);
    if (w9_ok(d)) return w9_splits(d);
    if (w9s_ok(d)) return w9s_splits(d);
    if (wgrad_use_v2(d)) {                                         // one 8-wave block per CU: ~256 blocks in all
        int64_t tiles = (int64_t)((d->R * d->S * d->C + 255) / 256) * ((d->K + 255) / 256);
        int64_t sp = 256 / tiles, maxs = (P + 127) / 128;
        if (sp > maxs) sp = maxs;
        if (sp < 1) sp = 1;
        if (sp > 64) sp = 64;
        return (int)sp;
    }
    int bnw = d->K >= 128 ? 128 : (d->K > 16 ? 64 : 16);
    int64_t tiles = (int64_t)((d->R * d->S * d->C + 127) / 128) * ((d->K + bnw - 1) / bnw);
    int64_t want = (1024 + tiles - 1) / tiles;                 // ~4 blocks per CU
    int64_t maxs = (P + 255) / 256;                            // >= 256 pixels per split
    int64_t sp = want < maxs ? want : maxs;
    int64_t slab = (int64_t)d->R * d->S * d->C * d->K * 4;     // bytes per split
    int64_t cap = slab <= (1 << 20) ? 256 : 64;                // small slabs: the reduce pass stays cheap
    if (sp < 1) sp = 1;
    if (sp > cap) sp = cap;
    return (int)sp;
}

struct W9Net { const void* x; const void* dy; const void* x2; const void* dy2; float* dw; };
// one network (nb == nullptr), or two networks of the same layer shape sharing the launch: each gets half the splits
static int run_w9(const sgg_conv_desc* d, const W9Net& na, const W9Net* nb, int Cr, int Kr, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
    const int sp = w9_splits(d);
    if (nb && (sp & 1)) return SGG_EUNSUPPORTED;
    size_t need9 = (size_t)sp * 9 * d->C * d->K * sizeof(float);
    if (ws_bytes < need9 || !ws) return SGG_EWORKSPACE;
    const int spn = nb ? sp / 2 : sp;                    // splits per network
    W9Args w;
    w.x = (const char*)na.x; w.dy = (const char*)na.dy; w.x2 = (const char*)na.x2; w.dy2 = (const char*)na.dy2; w.ws = (float*)ws;
    w.xb = w.dyb = w.x2b = w.dy2b = nullptr;
    if (nb) { w.xb = (const char*)nb->x; w.dyb = (const char*)nb->dy; w.x2b = (const char*)nb->x2; w.dy2b = (const char*)nb->dy2; }
    w.splits_per_net = spn;
    w.N = d->N; w.H = d->H; w.W = d->W; w.C = d->C; w.K = d->K; w.reflect = d->pad_mode == SGG_PAD_REFLECT;
    w.tiles = w9_tiles(d) * (na.x2 ? 2 : 1);
    w.tiles_per_split = (w.tiles + spn - 1) / spn;
    SGG_LDS_ATTR(conv3x3_wgrad_halo_kernel, 2 * W9_STAGE);
    sgg_launch_timed(conv3x3_wgrad_halo_kernel, dim3((unsigned)(sp * (d->C / 64) * (d->K / 128))), dim3(512), (unsigned)(2 * W9_STAGE), s, w);
    int rc9 = sgg_check_launch();
    if (rc9) return rc9;
    int64_t total9 = (int64_t)9 * Cr * (d->K / 4) * (nb ? 2 : 1);
    int blocks9 = (int)((total9 + 255) / 256); if (blocks9 > 4096) blocks9 = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks9), dim3(256), 0, s, (const float*)ws, na.dw, 9, d->C, d->K, Cr, Kr, spn, accumulate,
                       nb ? nb->dw : (float*)nullptr);
    return sgg_check_launch();
}

template <typename T>
// (xb, dyb, dwb): the same call site of a SECOND network (sgg_*_bwd_weight_group2): its main kernel runs right after the first one's
// into the slabs behind them, and ONE reduce launch sums both sets (each in the single call's order: bit-identical results).
static int run_wgrad(const sgg_conv_desc* d, const void* x, const void* dy, float* dw, int Cr, int Kr, int accumulate, void* ws, size_t ws_bytes, hipStream_t s,
                     const void* xb = nullptr, const void* dyb = nullptr, float* dwb = nullptr) {
    const int nets = xb ? 2 : 1;
    WgradArgs a;
    a.x = (const char*)x; a.dy = (const char*)dy; a.ws = (float*)ws;
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.K = d->K; a.R = d->R; a.S = d->S; a.stride = d->stride;
    a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.Ho = d->Ho; a.Wo = d->Wo; a.reflect = d->pad_mode == SGG_PAD_REFLECT;
    a.P = d->N * d->Ho * d->Wo;
    a.xb = a.x; a.dyb = a.dy; a.splits_per_net = 1 << 30;
    if constexpr (sizeof(T) == 2) {                     // the two 7x7 layers with a 3-channel side
        if (use_glds() && (w7_head_ok(d) || w7_stem_ok(d))) {
            const bool stem = !w7_head_ok(d);
            int rc7 = run_w7(d, stem, x, dy, dw, Cr, Kr, accumulate, ws, ws_bytes, s);
            return (rc7 || !xb) ? rc7 : run_w7(d, stem, xb, dyb, dwb, Cr, Kr, accumulate, ws, ws_bytes, s);
        }
    }
    if (halo_wgrad_ok(d)) {
        const int nb = halo_wgrad_blocks(d), ntiles = d->N * (d->H / HALO_TH) * (d->W / HALO_TW);
        size_t need = (size_t)nb * d->R * d->S * d->C * d->K * sizeof(float);
        if (ws_bytes < need * nets || !ws) return SGG_EWORKSPACE;
        a.pix_per_split = 0; a.dHW = make_fastdiv(1); a.dW = make_fastdiv(1);
        size_t lds = 512 * 16 * sizeof(T) + (size_t)(HALO_TH + d->R - 1) * (HALO_TW + d->S - 1) * 128 + 1024;
        auto kern = conv_halo_wgrad_kernel<T>;
        SGG_LDS_ATTR(kern, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, s, a, ntiles);
        if (xb) {
            a.x = (const char*)xb; a.dy = (const char*)dyb; a.ws = (float*)((char*)ws + need);
            hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, s, a, ntiles);
        }
        int rc0 = sgg_check_launch();
        if (rc0) return rc0;
        int64_t total0 = (int64_t)d->R * d->S * Cr * (d->K / 4) * nets;
        int blocks0 = (int)((total0 + 255) / 256); if (blocks0 > 4096) blocks0 = 4096;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks0), dim3(256), 0, s, (const float*)ws, dw, d->R * d->S, d->C, d->K, Cr, Kr, nb, accumulate, dwb);
        return sgg_check_launch();
    }
    if constexpr (sizeof(T) == 2) {
        if (w9_ok(d)) {                                   // 3x3 s1: x halo resident, all taps per block
            if (!xb) return run_w9(d, W9Net{x, dy, nullptr, nullptr, dw}, nullptr, Cr, Kr, accumulate, ws, ws_bytes, s);
            // two networks (grouped call): one launch, half the blocks and half the split slabs per network -- as for the stride-2
            // shapes below, the slabs are the kernel's main traffic (D.h3: 75 MB written + read per network); the result equals
            // two single calls up to f32 summation order
            const W9Net second{xb, dyb, nullptr, nullptr, dwb};
            if (!(w9_splits(d) & 1)) return run_w9(d, W9Net{x, dy, nullptr, nullptr, dw}, &second, Cr, Kr, accumulate, ws, ws_bytes, s);
            int rc9 = run_w9(d, W9Net{x, dy, nullptr, nullptr, dw}, nullptr, Cr, Kr, accumulate, ws, ws_bytes, s);
            return rc9 ? rc9 : run_w9(d, second, nullptr, Cr, Kr, accumulate, ws, ws_bytes, s);
        }
        if (w9s_ok(d)) {                                  // 3x3 s2: the same with a parity-de-interleaved halo
            // Two networks (grouped call): ONE launch of the single call's grid in which each network gets half the blocks, i.e.
            // half the slabs.  The split slabs are this kernel's main cost next to the operands -- 75 MB written and 75 MB read
            // again per network at the generator's four stride-2 layers, for 67 MB of x and dy -- so the two-launch form of the
            // grouped call paid them twice.  Half the slabs = another f32 summation order than the single call's: for these
            // shapes the grouped result equals two single calls up to that order (1e-6 relative), not bit for bit.
            int sp = w9s_splits(d);
            const bool merged = xb && sp >= 2;
            const int spn = merged ? sp / 2 : sp;                  // slabs per network
            if (merged) sp = 2 * spn;
            size_t need9 = (size_t)spn * 9 * d->C * d->K * sizeof(float);
            if (ws_bytes < need9 * nets || !ws) return SGG_EWORKSPACE;
            W9SArgs w;
            w.x = (const char*)x; w.dy = (const char*)dy; w.ws = (float*)ws;
            w.xb = (const char*)(merged ? xb : x); w.dyb = (const char*)(merged ? dyb : dy); w.splits_per_net = merged ? spn : (1 << 30);
            w.N = d->N; w.H = d->H; w.W = d->W; w.C = d->C; w.K = d->K; w.Ho = d->Ho; w.Wo = d->Wo; w.pad_t = d->pad_t; w.pad_l = d->pad_l;
            w.tiles = w9s_tiles(d); w.tiles_per_split = (w.tiles + spn - 1) / spn;
            SGG_LDS_ATTR(conv3x3_wgrad_halo_s2_kernel, 2 * W9S_STAGE);
            hipLaunchKernelGGL(conv3x3_wgrad_halo_s2_kernel, dim3((unsigned)(sp * (d->C / 64) * (d->K / 128))), dim3(512), 2 * W9S_STAGE, s, w);
            if (xb && !merged) {
                w.x = (const char*)xb; w.dy = (const char*)dyb; w.ws = (float*)((char*)ws + need9);
                hipLaunchKernelGGL(conv3x3_wgrad_halo_s2_kernel, dim3((unsigned)(sp * (d->C / 64) * (d->K / 128))), dim3(512), 2 * W9S_STAGE, s, w);
            }
            int rc9 = sgg_check_launch();
            if (rc9) return rc9;
            int64_t total9 = (int64_t)9 * Cr * (d->K / 4) * nets;
            int blocks9 = (int)((total9 + 255) / 256); if (blocks9 > 4096) blocks9 = 4096;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks9), dim3(256), 0, s, (const float*)ws, dw, 9, d->C, d->K, Cr, Kr, spn, accumulate, dwb);
            return sgg_check_launch();
        }
    }
    int splits = wgrad_splits(d);
    const bool s2n = sizeof(T) == 2 && s2n_wgrad_ok(d) && Cr <= 4;   // D.h0: the vector-ALU kernel, one slab per block
    // grouped call of the LDS-DMA kernel: one launch, half the splits (= slabs: 8 MB each at D.h31) per network -- see the stride-2
    // halo branch above; equal to two single calls up to f32 summation order
    const bool merged = xb && !s2n && wgrad_use_v2(d) && splits >= 2;
    if (merged) splits /= 2;
    a.pix_per_split = (int)align_up((size_t)((a.P + splits - 1) / splits), 64);
    if (!s2n) splits = (a.P + a.pix_per_split - 1) / a.pix_per_split;
    a.dHW = make_fastdiv((uint32_t)(d->Ho * d->Wo)); a.dW = make_fastdiv((uint32_t)d->Wo);
    size_t need = (size_t)splits * d->R * d->S * d->C * d->K * sizeof(float);
    if (ws_bytes < need * nets || !ws) return SGG_EWORKSPACE;
    int rc = SGG_OK;
    if (merged) { a.xb = (const char*)xb; a.dyb = (const char*)dyb; a.splits_per_net = splits; }
    for (int net = 0; net < (merged ? 1 : nets) && rc == SGG_OK; ++net) {
    if (net) { a.x = (const char*)xb; a.dy = (const char*)dyb; a.ws = (float*)((char*)ws + need); }
    if (s2n) rc = launch_s2n_wgrad(d, a.x, a.dy, a.ws, Cr, s);
    else if (wgrad_use_v2(d)) {
        constexpr size_t lds = 2 * 2 * (size_t)(sizeof(T) == 2 ? 64 : 32) * 256 * sizeof(T);
        SGG_LDS_ATTR((conv_wgrad_glds_kernel<T, false>), lds);
        SGG_LDS_ATTR((conv_wgrad_glds_kernel<T, true>), lds);
        dim3 grid((unsigned)(((d->R * d->S * d->C + 255) / 256) * ((d->K + 255) / 256) * splits * (merged ? 2 : 1)));
        const int bkp = sizeof(T) == 2 ? 64 : 32;                      // pixels per stage (BKP in the kernel)
        const int rf = sgg_config().wgrad_rowfast;
        if (rf && d->Wo % bkp == 0) hipLaunchKernelGGL((conv_wgrad_glds_kernel<T, true>), grid, dim3(512), lds, s, a);
        else hipLaunchKernelGGL((conv_wgrad_glds_kernel<T, false>), grid, dim3(512), lds, s, a);
        rc = sgg_check_launch();
    } else if (d->K >= 128) rc = launch_wgrad_cfg<T, 128, 128, 2>(a, splits, s);
    else if (d->K > 16) rc = launch_wgrad_cfg<T, 128, 64, 4>(a, splits, s);
    else rc = launch_wgrad_cfg<T, 128, 16, 4>(a, splits, s);
    }
    if (rc) return rc;
    if (s2n) {
        const int outs = d->R * d->S * Cr * (d->K / 4);
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)((outs + 3) / 4), (unsigned)nets), dim3(256), 0, s, (const float*)ws, dw, dwb,
                           d->R * d->S, d->C, d->K, Cr, Kr, splits, accumulate);
        return sgg_check_launch();
    }
    int64_t total = (int64_t)d->R * d->S * Cr * (d->K / 4) * nets;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)ws, dw, d->R * d->S, d->C, d->K, Cr, Kr, splits, accumulate, dwb);
    return sgg_check_launch();
}

extern "C" {

#ifdef SGG_LAB
// lab build only: the per-phase stamps of SGG_ABLATE=8 (768 values)
int sgg_debug_phases(unsigned long long* out768) {
    if (!out768) return SGG_EINVAL;
    return hipMemcpyFromSymbol(out768, HIP_SYMBOL(g_dbg_phase), sizeof(g_dbg_phase)) == hipSuccess ? SGG_OK : SGG_ELAUNCH;
}
// lab build only (not part of include/sggan.h): timestamps left by the halo GEMM under SGG_ABLATE=9 and the wall-clock rate (kHz)
int sgg_debug_clocks(unsigned long long* out5) {
    if (!out5) return SGG_EINVAL;
    if (hipMemcpyFromSymbol(out5, HIP_SYMBOL(g_dbg_clk), 4 * sizeof(unsigned long long)) != hipSuccess) return SGG_ELAUNCH;
    int khz = 0, dev = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
    out5[4] = (unsigned long long)khz;
    return SGG_OK;
}

// debug aid (not part of include/sggan.h): occupancy the runtime reports for the hot kernels
int sgg_debug_occupancy(int* out, int cap) {
    int n = 0, v = 0;
    if (cap < 4) return SGG_EINVAL;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, conv_gemm_glds_kernel<bf16, MODE_FWD, 256, 256, 2, 8, 128, 2>, 512, 131072); out[n++] = v;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, conv_gemm_glds_kernel<bf16, MODE_DGRAD, 256, 256, 2, 8, 128, 2>, 512, 131072); out[n++] = v;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, conv_gemm_kernel<bf16, MODE_FWD, 128, 128, 2>, 256, 65536); out[n++] = v;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, conv_wgrad_kernel<bf16, 128, 128, 2>, 256, 32768); out[n++] = v;
    return n;
}
#endif  // SGG_LAB

int sgg_pack_conv_weights(const float* w, int R, int S, int C, int K, int Cpad, int Kpad, int dtype, void* wf, void* wd, void* stream) {
    if (!w || R <= 0 || S <= 0 || C <= 0 || K <= 0 || Cpad < C || Kpad < K || Cpad % SGG_CPAD || Kpad % SGG_CPAD) return SGG_EINVAL;
    int64_t total = (int64_t)R * S * Cpad * Kpad;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    if (dtype == SGG_BF16) hipLaunchKernelGGL(pack_weights_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, R * S, C, K, Cpad, Kpad, (bf16*)wf, (bf16*)wd);
    else if (dtype == SGG_F32) hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, R * S, C, K, Cpad, Kpad, (float*)wf, (float*)wd);
    else return SGG_EINVAL;
    return sgg_check_launch();
}

int sgg_pack_conv_weights_batch(const sgg_pack_item* items_dev, int n_items, int64_t max_elems, int dtype, void* stream) {
    if (!items_dev || n_items <= 0 || max_elems <= 0) return SGG_EINVAL;
    int blocks = (int)((max_elems + 4095) / 4096); if (blocks > 1024) blocks = 1024;     // 64 x 64 tiles, grid-strided
    dim3 grid((unsigned)blocks, (unsigned)n_items);
    if (dtype == SGG_BF16) hipLaunchKernelGGL(pack_weights_batch_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, items_dev);
    else if (dtype == SGG_F32) hipLaunchKernelGGL(pack_weights_batch_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, items_dev);
    else return SGG_EINVAL;
    return sgg_check_launch();
}

size_t sgg_conv2d_fwd_workspace(const sgg_conv_desc* d) {
    return desc_ok(d) ? plan_gemm(d, MODE_FWD).ws_bytes : 0;
}

int sgg_conv2d_fwd(const sgg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act, float leak,
                   void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || !x || !w || !y) return SGG_EINVAL;
    ConvArgs a = make_args(d, x, w, bias, y, act, leak);
    if (halo_narrow_in_ok(d, d->C, d->K))               // narrow input (stem): halo + whole weight matrix resident in LDS
        return d->dtype == SGG_BF16 ? launch_halo_narrow_in<bf16>(d, a, 0, (hipStream_t)stream) : launch_halo_narrow_in<float>(d, a, 0, (hipStream_t)stream);
    if (use_glds() && n7_fwd_ok(d)) {                    // the 7x7 head: (channel, tap column) in the GEMM N dimension
        N7Args q;
        q.src = (const char*)x; q.wmat = (const char*)w; q.bias = bias; q.dst = y;
        q.N = d->N; q.H = d->H; q.W = d->W; q.Ho = d->Ho; q.Wo = d->Wo; q.pt = 3; q.pl = 3; q.K = 3;
        q.reflect = d->pad_mode == SGG_PAD_REFLECT; q.flip = 0; q.act = act; q.leak = leak; q.dst_f32 = 0;
        return launch_n7(q, (hipStream_t)stream);
    }
    if (use_glds() && halo_fwd_ok(d))                   // narrow output at full resolution: input halo resident in LDS
        return d->dtype == SGG_BF16 ? launch_halo_fwd<bf16>(d, a, (hipStream_t)stream) : launch_halo_fwd<float>(d, a, (hipStream_t)stream);
    return d->dtype == SGG_BF16 ? run_gemm<bf16, MODE_FWD>(d, a, ws, ws_bytes, (hipStream_t)stream)
                                : run_gemm<float, MODE_FWD>(d, a, ws, ws_bytes, (hipStream_t)stream);
}

// Pixel chunks per image of the (sum, sumsq) rows sgg_conv2d_fwd_stats() emits; 0 = this shape has no such epilogue
static size_t halo3_fwd_stats_chunks(const sgg_conv_desc* d) {           // the LDS-resident 3x3 kernel: a row per (2 x 128 tile, tile row)
    if (!desc_ok(d) || d->dtype != SGG_BF16) return 0;
    ConvArgs a = make_args(d, nullptr, nullptr, nullptr, nullptr, SGG_ACT_NONE, 0.f);
    if (plan_gemm(d, MODE_FWD).ksplit > 1 || !halo3_ok(a, MODE_FWD, true)) return 0;
    return (size_t)(d->H / 2) * (d->W / H3_TW) * 2;
}
size_t sgg_conv2d_fwd_stats_chunks(const sgg_conv_desc* d) {
    if (!desc_ok(d) || d->dtype != SGG_BF16) return 0;
    if (NI_WIDE && halo_narrow_in_ok(d, d->C, d->K)) return (size_t)(d->H / HALO_TH) * (d->W / HALO_TW);   // the stem: a row per 16 x 32 tile
    return halo3_fwd_stats_chunks(d);
}

int sgg_conv2d_fwd_stats(const sgg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, float* partial,
                         void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || !x || !w || !y || !partial) return SGG_EINVAL;
    if (sgg_conv2d_fwd_stats_chunks(d) == 0) return SGG_EUNSUPPORTED;
    ConvArgs a = make_args(d, x, w, bias, y, SGG_ACT_NONE, 0.f);
    a.stats = partial;
    if (halo_narrow_in_ok(d, d->C, d->K)) return launch_halo_narrow_in<bf16>(d, a, 0, (hipStream_t)stream);
    return run_gemm<bf16, MODE_FWD>(d, a, ws, ws_bytes, (hipStream_t)stream);
}

static size_t fold_bytes(const sgg_conv_desc* d) {
    if (d->pad_mode != SGG_PAD_REFLECT || !use_glds()) return 0;
    size_t es = d->dtype == SGG_BF16 ? 2 : 4;
    return align_up((size_t)d->N * fold_border_per_image(d->H, d->W, d->pad_t) * d->R * d->S * d->K * es, 256);
}

size_t sgg_conv2d_bwd_data_workspace(const sgg_conv_desc* d) {
    if (!desc_ok(d)) return 0;
    if (use_glds() && n7_dgrad_ok(d)) return n7_dgrad_ws(d);
    return fold_bytes(d) + plan_gemm(d, MODE_DGRAD).ws_bytes;
}

struct NormBwdStats { const void* nx; const float* nstats; const float* ngamma; const float* nbeta; int nact; float nleak; float* partial; int mixed; const void* w2; int nsplit; };

static int conv2d_bwd_data_impl(const sgg_conv_desc* d, const void* dy, const void* w, const void* addend, void* dx, const NormBwdStats* nb,
                                void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || !dy || !w || !dx) return SGG_EINVAL;
    ConvArgs a = make_args(d, dy, w, nullptr, dx, SGG_ACT_NONE, 0.f);
    a.addend = (const char*)addend;
    if (nb && nb->w2) { a.wmat2 = (const char*)nb->w2; a.nsplit = nb->nsplit; }
    if (nb && nb->mixed) { a.dst_f32 = nb->mixed & 1; a.addend_f32 = (nb->mixed >> 1) & 1; }
    if (nb && nb->partial) {
        a.stats = nb->partial; a.nx = (const char*)nb->nx; a.nstats = nb->nstats; a.ngamma = nb->ngamma; a.nbeta = nb->nbeta;
        a.nact = nb->nact; a.nleak = nb->nleak;
    } else nb = nullptr;                                 // from here on `nb` means: norm-backward sums wanted
    if (!addend && halo_narrow_in_ok(d, d->K, d->C)) {  // data-gradient of a narrow-OUTPUT conv (the head): dy has 8 channels
        int rc0 = d->dtype == SGG_BF16 ? launch_halo_narrow_in<bf16>(d, a, 1, (hipStream_t)stream) : launch_halo_narrow_in<float>(d, a, 1, (hipStream_t)stream);
        if (rc0 || !a.reflect) return rc0;
        // REFLECT: add the mirrored (MirrorPadGrad) terms of the border pixels with the small register-path launch
        return d->dtype == SGG_BF16 ? launch_gemm<bf16, MODE_BORDER>(a, (hipStream_t)stream) : launch_gemm<float, MODE_BORDER>(a, (hipStream_t)stream);
    }
    if (!nb && s2n_dgrad_ok(d)) return launch_s2n_dgrad(d, dy, w, nullptr, 0, addend, dx, (hipStream_t)stream);
    if (!nb && use_glds() && n7_dgrad_ok(d)) {            // (addend: joined in the fold pass, before its one rounding)
        // the stem: data gradient on the PADDED grid (a zero-padded "full" correlation of dy with the mirrored taps, f32),
        // then MirrorPadGrad as a fold of the 3-pixel frame onto the image -- one rounding, no separate border GEMM
        if (!ws || ws_bytes < n7_dgrad_ws(d)) return SGG_EWORKSPACE;
        N7Args q;
        q.src = (const char*)dy; q.wmat = (const char*)w; q.bias = nullptr; q.dst = ws;
        q.N = d->N; q.H = d->H; q.W = d->W; q.Ho = d->H + 6; q.Wo = d->W + 6; q.pt = 6; q.pl = 6; q.K = 3;
        q.reflect = 0; q.flip = 1; q.act = SGG_ACT_NONE; q.leak = 0.f; q.dst_f32 = 1;
        int rc0 = launch_n7(q, (hipStream_t)stream);
        if (rc0) return rc0;
        const int64_t total = (int64_t)d->N * d->H * d->W;
        int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(pad3_fold_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)ws, (const char*)addend, (char*)dx, d->N, d->H, d->W);
        return sgg_check_launch();
    }
    if (!addend && halo_dgrad_narrow_ok(d)) {
        ConvArgs h = a;                                  // the same computation written as a forward conv over dy
        h.C = d->K; h.K = d->C; h.reflect = 0;
        int rc0 = d->dtype == SGG_BF16 ? launch_halo_fwd<bf16>(d, h, (hipStream_t)stream, 1) : launch_halo_fwd<float>(d, h, (hipStream_t)stream, 1);
        if (rc0 || !a.reflect) return rc0;
        return d->dtype == SGG_BF16 ? launch_gemm<bf16, MODE_BORDER>(a, (hipStream_t)stream) : launch_gemm<float, MODE_BORDER>(a, (hipStream_t)stream);
    }
    const size_t fb = fold_bytes(d);
    if (fb) {
        // v2: pre-fold the gather rows of the border pixels, then ONE GEMM launch reads them like any other source
        if (!ws || ws_bytes < fb) return SGG_EWORKSPACE;
        size_t es = d->dtype == SGG_BF16 ? 2 : 4;
        int64_t total = (int64_t)d->N * fold_border_per_image(d->H, d->W, d->pad_t) * d->R * d->S * d->K * es / 16;
        GemmPlan gp = plan_gemm(d, MODE_DGRAD);
        ConvArgs probe = a; probe.ksplit = gp.ksplit;
        const bool halo3 = halo3_ok(probe, MODE_DGRAD, d->dtype == SGG_BF16);
        if (halo3) total = ((int64_t)d->N * d->H * 18 + (int64_t)d->N * 2 * d->W) * d->K * es / 16;   // smaller side tensor
        int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
        if (halo3) hipLaunchKernelGGL(fold_halo_gather_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)dy, (char*)ws, d->N, d->H, d->W, d->K);
        else if (d->dtype == SGG_BF16) hipLaunchKernelGGL(fold_gather_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)dy, (char*)ws, d->N, d->H, d->W, d->K, d->R, d->S, d->pad_t, d->Ho, d->Wo);
        else hipLaunchKernelGGL(fold_gather_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)dy, (char*)ws, d->N, d->H, d->W, d->K, d->R, d->S, d->pad_t, d->Ho, d->Wo);
        a.fold = (const char*)ws;
    }
    void* ws2 = ws ? (char*)ws + fb : nullptr;
    size_t ws2_bytes = ws_bytes > fb ? ws_bytes - fb : 0;
    int rc = d->dtype == SGG_BF16 ? run_gemm<bf16, MODE_DGRAD>(d, a, ws2, ws2_bytes, (hipStream_t)stream)
                                  : run_gemm<float, MODE_DGRAD>(d, a, ws2, ws2_bytes, (hipStream_t)stream);
    if (rc || !a.reflect || use_glds()) return rc;
    // v1 REFLECT: a second, small launch adds the mirrored (MirrorPadGrad) terms to the border pixels
    return d->dtype == SGG_BF16 ? launch_gemm<bf16, MODE_BORDER>(a, (hipStream_t)stream) : launch_gemm<float, MODE_BORDER>(a, (hipStream_t)stream);
}

int sgg_conv2d_bwd_data(const sgg_conv_desc* d, const void* dy, const void* w, const void* addend, void* dx, void* ws, size_t ws_bytes, void* stream) {
    return conv2d_bwd_data_impl(d, dy, w, addend, dx, nullptr, ws, ws_bytes, stream);
}

// Mixed-precision data gradient (bf16 operands, f32 result): only the LDS-resident 3x3 halo GEMM has the epilogue
int sgg_conv2d_bwd_data_mixed_supported(const sgg_conv_desc* d) {
    if (!desc_ok(d) || d->dtype != SGG_BF16) return 0;
    ConvArgs a = make_args(d, nullptr, nullptr, nullptr, nullptr, SGG_ACT_NONE, 0.f);
    return (plan_gemm(d, MODE_DGRAD).ksplit == 1 && halo3_ok(a, MODE_DGRAD, true)) ? 1 : 0;
}
int sgg_conv2d_bwd_data_mixed(const sgg_conv_desc* d, const void* dy, const void* w, const void* addend, int addend_is_f32, float* dx,
                              void* ws, size_t ws_bytes, void* stream) {
    if (!sgg_conv2d_bwd_data_mixed_supported(d)) return SGG_EUNSUPPORTED;
    NormBwdStats flags{};                            // nb == nullptr semantics, but carries the two mixed flags
    flags.mixed = 1 | (addend_is_f32 ? 2 : 0);
    return conv2d_bwd_data_impl(d, dy, w, addend, dx, &flags, ws, ws_bytes, stream);
}

// Pixel chunks per image of the instance-norm-backward partial sums sgg_conv2d_bwd_data_stats() emits; 0 = unsupported shape
size_t sgg_conv2d_bwd_data_stats_chunks(const sgg_conv_desc* d) {
    if (!desc_ok(d) || d->dtype != SGG_BF16) return 0;
    ConvArgs a = make_args(d, nullptr, nullptr, nullptr, nullptr, SGG_ACT_NONE, 0.f);
    if (plan_gemm(d, MODE_DGRAD).ksplit > 1 || !halo3_ok(a, MODE_DGRAD, true)) return 0;
    return (size_t)(d->H / 2) * (d->W / H3_TW) * 2;
}

int sgg_conv2d_bwd_data_stats(const sgg_conv_desc* d, const void* dy, const void* w, const void* addend, void* dx,
                              const void* norm_x, const float* norm_stats, const float* norm_gamma, const float* norm_beta,
                              int norm_act, float norm_leak, float* partial, void* ws, size_t ws_bytes, void* stream) {
    if (!norm_x || !norm_stats || !norm_gamma || !norm_beta || !partial) return SGG_EINVAL;
    if (norm_act == SGG_ACT_TANH) return SGG_EUNSUPPORTED;
    if (sgg_conv2d_bwd_data_stats_chunks(d) == 0) return SGG_EUNSUPPORTED;
    NormBwdStats nb{norm_x, norm_stats, norm_gamma, norm_beta, norm_act, norm_leak, partial, 0, nullptr, 0};
    return conv2d_bwd_data_impl(d, dy, w, addend, dx, &nb, ws, ws_bytes, stream);
}

// Two networks of one shape on a stacked batch (images >= nsplit use the second weight set): one launch of the LDS-resident
// 3x3 kernels instead of two -- 512 blocks, so a CU's second block loads its halo while the first one's stores drain.
int sgg_conv2d_pair_supported(const sgg_conv_desc* d) {
    if (!desc_ok(d) || d->dtype != SGG_BF16 || d->N < 2) return 0;
    ConvArgs a = make_args(d, nullptr, nullptr, nullptr, nullptr, SGG_ACT_NONE, 0.f);
    return (plan_gemm(d, MODE_FWD).ksplit == 1 && plan_gemm(d, MODE_DGRAD).ksplit == 1 && halo3_ok(a, MODE_FWD, true) && halo3_ok(a, MODE_DGRAD, true)) ? 1 : 0;
}
int sgg_conv2d_fwd_stats_pair(const sgg_conv_desc* d, const void* x, const void* w, const float* bias, const void* w2, const float* bias2,
                              int nsplit, void* y, float* partial, void* ws, size_t ws_bytes, void* stream) {
    if (!sgg_conv2d_pair_supported(d)) return SGG_EUNSUPPORTED;
    if (!x || !w || !w2 || !y || !partial || nsplit <= 0 || nsplit >= d->N) return SGG_EINVAL;
    ConvArgs a = make_args(d, x, w, bias, y, SGG_ACT_NONE, 0.f);
    a.stats = partial; a.wmat2 = (const char*)w2; a.bias2 = bias2; a.nsplit = nsplit;
    return run_gemm<bf16, MODE_FWD>(d, a, ws, ws_bytes, (hipStream_t)stream);
}
// "Normalise on load": conv(relu(instnorm(x_raw))) without the norm's apply pass -- x_raw is the previous conv's raw output,
// x_stats its (mean, rstd) (sgg_instnorm_finalize), and the normalised tensor comes out in x_norm as a by-product.
int sgg_conv2d_fwd_normload_supported(const sgg_conv_desc* d) {
    if (!desc_ok(d) || d->dtype != SGG_BF16 || d->pad_mode != SGG_PAD_REFLECT || d->C > H3_NORM_MAXC) return 0;
    return halo3_fwd_stats_chunks(d) != 0;
}
int sgg_conv2d_fwd_stats_normload(const sgg_conv_desc* d, const void* x_raw, const float* x_stats, const float* x_gamma, const float* x_beta,
                                  const float* x_gamma2, const float* x_beta2, void* x_norm, const void* w, const float* bias,
                                  const void* w2, const float* bias2, int nsplit, void* y, float* partial,
                                  void* ws, size_t ws_bytes, void* stream) {
    if (!sgg_conv2d_fwd_normload_supported(d)) return SGG_EUNSUPPORTED;
    if (!x_raw || !x_stats || !x_gamma || !x_beta || !x_norm || !w || !y || !partial) return SGG_EINVAL;
    ConvArgs a = make_args(d, x_raw, w, bias, y, SGG_ACT_NONE, 0.f);
    a.stats = partial; a.nstats = x_stats; a.ngamma = x_gamma; a.nbeta = x_beta; a.nout = (char*)x_norm;
    if (w2) {
        if (!sgg_conv2d_pair_supported(d) || !x_gamma2 || !x_beta2 || nsplit <= 0 || nsplit >= d->N) return SGG_EINVAL;
        a.wmat2 = (const char*)w2; a.bias2 = bias2; a.nsplit = nsplit; a.ngamma2 = x_gamma2; a.nbeta2 = x_beta2;
    }
    return run_gemm<bf16, MODE_FWD>(d, a, ws, ws_bytes, (hipStream_t)stream);
}
int sgg_conv2d_bwd_data_pair(const sgg_conv_desc* d, const void* dy, const void* w, const void* w2, int nsplit, const void* addend, void* dx,
                             void* ws, size_t ws_bytes, void* stream) {
    if (!sgg_conv2d_pair_supported(d)) return SGG_EUNSUPPORTED;
    if (!w2 || nsplit <= 0 || nsplit >= d->N) return SGG_EINVAL;
    NormBwdStats flags{};
    flags.w2 = w2; flags.nsplit = nsplit;
    return conv2d_bwd_data_impl(d, dy, w, addend, dx, &flags, ws, ws_bytes, stream);
}

// ---- grouped launches: the same call site of TWO networks of one architecture (G_A->B beside G_B->A, D_A beside D_B).  `d`
// describes ONE network's call (N images); x / y (dy / dx, addend) hold 2N images, the first network's first; the result is
// exactly that of two single calls (bit for bit: each group runs the single call's tiles and split-K), in ONE launch for the
// generic GEMM and the stacked-batch halo kernels -- two half-size launches of the discriminators' small maps are latency
// bound -- and as two launches for the special 7x7 / narrow kernels.  ws >= 2 x the single call's workspace.
static inline size_t tensor_bytes(const sgg_conv_desc* d, bool out_side) {
    const size_t es = d->dtype == SGG_BF16 ? 2 : 4;
    return out_side ? (size_t)d->N * d->Ho * d->Wo * d->K * es : (size_t)d->N * d->H * d->W * d->C * es;
}
int sgg_conv2d_fwd_group2(const sgg_conv_desc* d, const void* x, const void* w, const float* bias, const void* w2, const float* bias2,
                          void* y, int act, float leak, void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || !x || !w || !w2 || !y) return SGG_EINVAL;
    const size_t xin = tensor_bytes(d, false), yout = tensor_bytes(d, true);
    if (!use_glds() || halo_narrow_in_ok(d, d->C, d->K) || n7_fwd_ok(d) || halo_fwd_ok(d)) {
        int rc = sgg_conv2d_fwd(d, x, w, bias, y, act, leak, ws, ws_bytes, stream);
        return rc ? rc : sgg_conv2d_fwd(d, (const char*)x + xin, w2, bias2, (char*)y + yout, act, leak, ws, ws_bytes, stream);
    }
    ConvArgs a = make_args(d, x, w, bias, y, act, leak);
    a.grp = 2; a.wmat2 = (const char*)w2; a.bias2 = bias2; a.net_src = xin; a.net_dst = yout;
    return d->dtype == SGG_BF16 ? run_gemm<bf16, MODE_FWD>(d, a, ws, ws_bytes, (hipStream_t)stream)
                                : run_gemm<float, MODE_FWD>(d, a, ws, ws_bytes, (hipStream_t)stream);
}
int sgg_conv2d_bwd_data_group2(const sgg_conv_desc* d, const void* dy, const void* w, const void* w2, const void* addend, void* dx,
                               void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || !dy || !w || !w2 || !dx) return SGG_EINVAL;
    const size_t xin = tensor_bytes(d, false), yout = tensor_bytes(d, true);
    if (s2n_dgrad_ok(d)) {                               // D.h0: one launch over the stacked batch, weights picked per image
        sgg_conv_desc d2 = *d;
        d2.N = 2 * d->N;
        return launch_s2n_dgrad(&d2, dy, w, w2, d->N, addend, dx, (hipStream_t)stream);
    }
    const bool special = !use_glds() || n7_dgrad_ok(d) || (!addend && (halo_narrow_in_ok(d, d->K, d->C) || halo_dgrad_narrow_ok(d))) || fold_bytes(d) > 0;
    if (special) {                                       // (REFLECT shapes of the 3x3 halo kernel: sgg_conv2d_bwd_data_pair)
        int rc = sgg_conv2d_bwd_data(d, dy, w, addend, dx, ws, ws_bytes, stream);
        return rc ? rc : sgg_conv2d_bwd_data(d, (const char*)dy + yout, w2, addend ? (const char*)addend + xin : nullptr, (char*)dx + xin, ws, ws_bytes, stream);
    }
    ConvArgs a = make_args(d, dy, w, nullptr, dx, SGG_ACT_NONE, 0.f);
    a.addend = (const char*)addend;
    a.grp = 2; a.wmat2 = (const char*)w2; a.net_src = yout; a.net_dst = xin; a.net_add = xin;
    return d->dtype == SGG_BF16 ? run_gemm<bf16, MODE_DGRAD>(d, a, ws, ws_bytes, (hipStream_t)stream)
                                : run_gemm<float, MODE_DGRAD>(d, a, ws, ws_bytes, (hipStream_t)stream);
}
// Conv2DTranspose: `d` is the equivalent forward conv of ONE network (sgg_deconv2d_fwd); x: (2N,Ho,Wo,K), y: (2N,H,W,C)
int sgg_deconv2d_fwd_group2(const sgg_conv_desc* d, const void* x, const void* w, const float* bias, const void* w2, const float* bias2,
                            void* y, int act, float leak, void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || d->pad_mode != SGG_PAD_ZERO || !x || !w || !w2 || !y) return SGG_EINVAL;
    ConvArgs a = make_args(d, x, w, bias, y, act, leak);
    if (!use_glds()) {
        int rc = sgg_deconv2d_fwd(d, x, w, bias, y, act, leak, ws, ws_bytes, stream);
        return rc ? rc : sgg_deconv2d_fwd(d, (const char*)x + tensor_bytes(d, true), w2, bias2, (char*)y + tensor_bytes(d, false), act, leak, ws, ws_bytes, stream);
    }
    a.grp = 2; a.wmat2 = (const char*)w2; a.bias2 = bias2; a.net_src = tensor_bytes(d, true); a.net_dst = tensor_bytes(d, false);
    return d->dtype == SGG_BF16 ? run_gemm<bf16, MODE_DGRAD>(d, a, ws, ws_bytes, (hipStream_t)stream)
                                : run_gemm<float, MODE_DGRAD>(d, a, ws, ws_bytes, (hipStream_t)stream);
}
int sgg_deconv2d_bwd_data_group2(const sgg_conv_desc* d, const void* dy, const void* w, const void* w2, void* dx,
                                 void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || d->pad_mode != SGG_PAD_ZERO || !dy || !w || !w2 || !dx) return SGG_EINVAL;
    if (!use_glds()) {
        int rc = sgg_deconv2d_bwd_data(d, dy, w, dx, ws, ws_bytes, stream);
        return rc ? rc : sgg_deconv2d_bwd_data(d, (const char*)dy + tensor_bytes(d, false), w2, (char*)dx + tensor_bytes(d, true), ws, ws_bytes, stream);
    }
    ConvArgs a = make_args(d, dy, w, nullptr, dx, SGG_ACT_NONE, 0.f);
    a.grp = 2; a.wmat2 = (const char*)w2; a.net_src = tensor_bytes(d, false); a.net_dst = tensor_bytes(d, true);
    return d->dtype == SGG_BF16 ? run_gemm<bf16, MODE_FWD>(d, a, ws, ws_bytes, (hipStream_t)stream)
                                : run_gemm<float, MODE_FWD>(d, a, ws, ws_bytes, (hipStream_t)stream);
}

size_t sgg_conv2d_bwd_weight_workspace(const sgg_conv_desc* d) {
    if (!desc_ok(d)) return 0;
    if (use_glds() && (w7_head_ok(d) || w7_stem_ok(d))) {
        const W7Plan p = w7_plan(d, w7_stem_ok(d));
        return p.slab_bytes + p.xpad_bytes;
    }
    return (size_t)wgrad_splits(d) * d->R * d->S * d->C * d->K * sizeof(float);
}

int sgg_conv2d_bwd_weight(const sgg_conv_desc* d, const void* x, const void* dy, float* dw, int Cr, int Kr, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || !x || !dy || !dw || Cr <= 0 || Kr <= 0 || Cr > d->C || Kr > d->K) return SGG_EINVAL;
    return d->dtype == SGG_BF16 ? run_wgrad<bf16>(d, x, dy, dw, Cr, Kr, accumulate, ws, ws_bytes, (hipStream_t)stream)
                                : run_wgrad<float>(d, x, dy, dw, Cr, Kr, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

// Grouped call (see sgg_conv2d_fwd_group2): x / dy hold 2N images, the first network's first; dw / dw2 are the two networks' gradients.
int sgg_conv2d_bwd_weight_group2(const sgg_conv_desc* d, const void* x, const void* dy, float* dw, float* dw2, int Cr, int Kr, int accumulate,
                                 void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || !x || !dy || !dw || !dw2 || Cr <= 0 || Kr <= 0 || Cr > d->C || Kr > d->K) return SGG_EINVAL;
    const char* xb = (const char*)x + tensor_bytes(d, false);
    const char* dyb = (const char*)dy + tensor_bytes(d, true);
    return d->dtype == SGG_BF16 ? run_wgrad<bf16>(d, x, dy, dw, Cr, Kr, accumulate, ws, ws_bytes, (hipStream_t)stream, xb, dyb, dw2)
                                : run_wgrad<float>(d, x, dy, dw, Cr, Kr, accumulate, ws, ws_bytes, (hipStream_t)stream, xb, dyb, dw2);
}

// Weight gradient of TWO applications of one layer (same shape) in one launch: dw (+)= wgrad(x0, dy0) + wgrad(x1, dy1).
// Only the shapes of the all-taps 3x3 kernel (SGG_EUNSUPPORTED otherwise: call sgg_conv2d_bwd_weight twice);
// workspace as for a single call.
int sgg_conv2d_bwd_weight_pair_supported(const sgg_conv_desc* d) { return desc_ok(d) && w9_ok(d) ? 1 : 0; }

int sgg_conv2d_bwd_weight_pair(const sgg_conv_desc* d, const void* x0, const void* dy0, const void* x1, const void* dy1, float* dw,
                               int Cr, int Kr, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || !x0 || !dy0 || !x1 || !dy1 || !dw || Cr <= 0 || Kr <= 0 || Cr > d->C || Kr > d->K) return SGG_EINVAL;
    if (!w9_ok(d)) return SGG_EUNSUPPORTED;
    return run_w9(d, W9Net{x0, dy0, x1, dy1, dw}, nullptr, Cr, Kr, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

// ... of TWO networks of one architecture (the cycle step's generators, each applied twice): four (x, dy) sets, two dW, one launch.
// Each network gets half the blocks, so half as many f32 slabs are written and reduced for the same work per CU; the sums differ
// from two sgg_conv2d_bwd_weight_pair calls only in f32 summation order (16 partial sums per network instead of 32).
int sgg_conv2d_bwd_weight_pair2(const sgg_conv_desc* d, const void* xa0, const void* dya0, const void* xa1, const void* dya1, float* dwa,
                                const void* xb0, const void* dyb0, const void* xb1, const void* dyb1, float* dwb,
                                int Cr, int Kr, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || !xa0 || !dya0 || !xa1 || !dya1 || !dwa || !xb0 || !dyb0 || !xb1 || !dyb1 || !dwb || Cr <= 0 || Kr <= 0 || Cr > d->C || Kr > d->K)
        return SGG_EINVAL;
    if (!w9_ok(d) || (w9_splits(d) & 1)) return SGG_EUNSUPPORTED;
    const W9Net nb{xb0, dyb0, xb1, dyb1, dwb};
    return run_w9(d, W9Net{xa0, dya0, xa1, dya1, dwa}, &nb, Cr, Kr, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

size_t sgg_deconv2d_fwd_workspace(const sgg_conv_desc* d) {
    return desc_ok(d) ? plan_gemm(d, MODE_DGRAD).ws_bytes : 0;
}
size_t sgg_deconv2d_bwd_data_workspace(const sgg_conv_desc* d) {
    return desc_ok(d) ? plan_gemm(d, MODE_FWD).ws_bytes : 0;
}

int sgg_deconv2d_fwd(const sgg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act, float leak,
                     void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || d->pad_mode != SGG_PAD_ZERO || !x || !w || !y) return SGG_EINVAL;
    ConvArgs a = make_args(d, x, w, bias, y, act, leak);
    return d->dtype == SGG_BF16 ? run_gemm<bf16, MODE_DGRAD>(d, a, ws, ws_bytes, (hipStream_t)stream)
                                : run_gemm<float, MODE_DGRAD>(d, a, ws, ws_bytes, (hipStream_t)stream);
}

// Conv2DTranspose forward that also emits the per-chunk (sum, sumsq) rows of its stored output for the instance norm behind it
// (module.py:254-260): one row per (image, 16 x 64 output-pixel tile) from the stride-2 halo kernel's epilogue.
size_t sgg_deconv2d_fwd_stats_chunks(const sgg_conv_desc* d) {
    if (!desc_ok(d) || d->dtype != SGG_BF16 || d->pad_mode != SGG_PAD_ZERO || !S2_WIDE) return 0;
    ConvArgs a = make_args(d, nullptr, nullptr, nullptr, nullptr, SGG_ACT_NONE, 0.f);
    if (!s2halo_ok(a, true)) return 0;
    return (size_t)(d->Ho / S2_TI) * (d->Wo / S2_TJ);
}
int sgg_deconv2d_fwd_stats(const sgg_conv_desc* d, const void* x, const void* w, const float* bias, const void* w2, const float* bias2, int nsplit,
                           void* y, float* partial, void* ws, size_t ws_bytes, void* stream) {
    (void)ws; (void)ws_bytes;
    if (!desc_ok(d) || !x || !w || !y || !partial) return SGG_EINVAL;
    if (sgg_deconv2d_fwd_stats_chunks(d) == 0) return SGG_EUNSUPPORTED;
    ConvArgs a = make_args(d, x, w, bias, y, SGG_ACT_NONE, 0.f);
    a.stats = partial;
    if (w2) {
        if (nsplit <= 0 || nsplit >= d->N) return SGG_EINVAL;
        a.wmat2 = (const char*)w2; a.bias2 = bias2; a.nsplit = nsplit;
    }
    return launch_s2halo(a, (hipStream_t)stream);
}

int sgg_deconv2d_bwd_data(const sgg_conv_desc* d, const void* dy, const void* w, void* dx, void* ws, size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || d->pad_mode != SGG_PAD_ZERO || !dy || !w || !dx) return SGG_EINVAL;
    ConvArgs a = make_args(d, dy, w, nullptr, dx, SGG_ACT_NONE, 0.f);
    return d->dtype == SGG_BF16 ? run_gemm<bf16, MODE_FWD>(d, a, ws, ws_bytes, (hipStream_t)stream)
                                : run_gemm<float, MODE_FWD>(d, a, ws, ws_bytes, (hipStream_t)stream);
}

int sgg_deconv2d_bwd_weight(const sgg_conv_desc* d, const void* x, const void* dy, float* dw, int Cr, int Kr, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    // deconv input x plays the conv-output role, deconv output-gradient dy the conv-input role
    return sgg_conv2d_bwd_weight(d, dy, x, dw, Cr, Kr, accumulate, ws, ws_bytes, stream);
}
int sgg_deconv2d_bwd_weight_group2(const sgg_conv_desc* d, const void* x, const void* dy, float* dw, float* dw2, int Cr, int Kr, int accumulate,
                                   void* ws, size_t ws_bytes, void* stream) {
    return sgg_conv2d_bwd_weight_group2(d, dy, x, dw, dw2, Cr, Kr, accumulate, ws, ws_bytes, stream);
}

}  // extern "C"
