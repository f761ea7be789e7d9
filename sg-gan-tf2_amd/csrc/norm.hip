// norm.hip -- instance normalisation (tfa.layers.InstanceNormalization, module.py:212..308; ops.py:13-22 by name)
// fused with the following activation (module.py:213,234,...; LeakyReLU :289-309) and the residual add (:217).
//
// HBM-bound.  NHWC: a thread owns one 16-byte channel vector and walks pixels, so every access is a coalesced
// 16-byte load and the per-(n,c) reduction never crosses channels.  Two passes per direction:
//   fwd : partial (sum, sumsq) per pixel-chunk -> finalize (mean, rstd; f64 combine, fixed order) -> apply
//   bwd : partial (sum g, sum g*xhat)          -> finalize (+ dgamma, dbeta over n)               -> apply
// Algorithmic bytes: fwd 2 reads + 1 write of the tensor, bwd 4 reads + 1 write.
#include "common.h"
#include <type_traits>

// Per-thread accumulators of the statistics passes.  On the f32 parity path they are f64: the backward sums (sum g,
// sum g*xhat) add a near-zero-mean field (the gradient that comes back through a SAME conv from another instance
// norm), so the sum cancels to ~1e-3 of its terms and an f32 running sum over 64+ pixels is off by 1e-4 of the result --
// visible against the float64 oracle (tools/diag_d_f32.py).  On the bf16 path the operands carry 8 bits: f32 is plenty.
// pixels per loop trip in the streaming passes (= 16-byte loads in flight per thread and tensor): 4 forward; 2 backward, where
// two tensors are read and 4 would cost occupancy (166 VGPRs: 3 waves/SIMD, measured 5-15 % slower than 2)
#define IN_U (BWD ? 2 : 4)
template <typename T> using InAcc = typename std::conditional<std::is_same<T, float>::value, double, float>::type;

// Two networks of the same shape run in lockstep on a batch that stacks their activations (images 0..nsplit-1 belong to
// the first, the rest to the second -- the cycle step's G_A->B / G_B->A and D_A / D_B pairs): the norm is per image, so one
// launch serves both and only the affine parameters (and their gradients) are picked by image index.  nsplit >= N: one network.
struct InSplit { const float* gamma2; const float* beta2; float* dgamma2; float* dbeta2; int nsplit; };
static inline InSplit in_nosplit() { InSplit s; s.gamma2 = nullptr; s.beta2 = nullptr; s.dgamma2 = nullptr; s.dbeta2 = nullptr; s.nsplit = 0x7fffffff; return s; }

// The incoming gradient dy normally has the tensor's storage type T.  Mixed mode (bf16 tensors, f32 gradient chain): dy is
// f32 -- the VEC channels of a 16-byte chunk of x are then two 16-byte chunks of dy.  `elem` = element index of the chunk.
#ifndef IN_NT
#define IN_NT 15               // streaming (nt) loads in the apply passes of tensors this pass reads for the last time before they leave the
                               // caches anyway: 1 forward x, 2 backward x, 4 backward dy, 8 the forward residual (A/B: profiles/r04_nt_loads.txt)
#endif
template <typename T, typename TG, bool NT = false>
__device__ inline void in_load_grad(const char* dy, size_t elem, float* gv) {
    if constexpr (std::is_same<T, TG>::value) {
        ET<T>::unpack(NT ? ld16_nt(dy + elem * sizeof(T)) : ld16(dy + elem * sizeof(T)), gv);
    } else {
        static_assert(std::is_same<TG, float>::value && ET<T>::VEC == 8, "mixed mode: f32 gradient of a bf16 tensor");
        ET<float>::unpack(ld16(dy + elem * 4), gv);
        ET<float>::unpack(ld16(dy + elem * 4 + 16), gv + 4);
    }
}

// pixels per partial-sum chunk: >= 64, and large enough that an image has at most ~128 chunks (the finalize kernels
// walk the chunks of one channel serially: 2048 chunks cost 68 us on a 256x512x64 tensor, 128 chunks 5 us)
static inline int in_rows_per_chunk(int64_t HW) {
    int64_t r = (HW + 127) / 128;
    r = (r + 63) / 64 * 64;
    return (int)(r < 64 ? 64 : (r > 4096 ? 4096 : r));
}
// ws layout: partial[N][chunks][C][2] f32, then sums[N][C][2] and tot[N][C][2] f32 (bwd only)
static inline int in_chunks(int64_t HW) { int r = in_rows_per_chunk(HW); return (int)((HW + r - 1) / r); }

template <typename T, bool BWD, typename TG = T>
__global__ __launch_bounds__(256) void in_partial_kernel(const char* x, const char* dy, const float* gamma, const float* beta,
                                                         const float* stats, float* partial, int64_t HW, int C, int chunks,
                                                         int rpc, int act, float leak, InSplit sp) {
    constexpr int VEC = ET<T>::VEC;
    const int CV = C / VEC;                       // channel vectors per pixel
    const int n = blockIdx.y, chunk = blockIdx.x;
    if (n >= sp.nsplit) { gamma = sp.gamma2; beta = sp.beta2; }
    const int64_t p0 = (int64_t)chunk * rpc;
    const int64_t p1 = p0 + rpc < HW ? p0 + rpc : HW;
    using AccT = InAcc<T>;
    __shared__ AccT red[256][2 * VEC + 1];
    // threads cover (pixel row, channel vector) pairs: cv = item % CV walks fastest
    for (int cvb = 0; cvb < CV; cvb += 256) {
        const int lanes = CV - cvb < 256 ? CV - cvb : 256;       // channel vectors handled in this sweep
        const int rows = 256 / lanes;                            // pixel rows in flight
        const int cv = cvb + (int)(threadIdx.x % lanes), prow = threadIdx.x / lanes;
        AccT s1[VEC], s2[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) s1[e] = s2[e] = (AccT)0;
        float mu[VEC], rs[VEC], gm[VEC], bt[VEC];
        if (BWD && prow < rows) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                int c = cv * VEC + e;
                mu[e] = stats[((size_t)n * C + c) * 2]; rs[e] = stats[((size_t)n * C + c) * 2 + 1];
                gm[e] = gamma[c]; bt[e] = beta[c];
            }
        }
        // IN_U pixels per trip: all their loads are issued before the first use (one 16-byte load in flight per thread left the
        // pass latency bound), and the activation code is resolved once -- the element loop is then branch-free selects.
        // The per-thread accumulation order (pixel order) is unchanged.
        if (prow < rows)
            act_dispatch(BWD ? act : SGG_ACT_NONE, [&](auto act_c) {
                constexpr int ACT = decltype(act_c)::value;
                auto accumulate = [&](const u32x4& xr, const float* gv) {
                    float xv[VEC];
                    ET<T>::unpack(xr, xv);
                    if (!BWD) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) { s1[e] += (AccT)xv[e]; s2[e] += (AccT)xv[e] * (AccT)xv[e]; }
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            float xh = (xv[e] - mu[e]) * rs[e];
                            float g = gv[e] * act_grad_c<ACT>(gm[e] * xh + bt[e], leak);
                            s1[e] += (AccT)g; s2[e] += (AccT)g * (AccT)xh;
                        }
                    }
                };
                const size_t base = ((size_t)n * HW * C + (size_t)cv * VEC) * sizeof(T), pstep = (size_t)C * sizeof(T);
                int64_t p = p0 + prow;
                for (; p + (IN_U - 1) * rows < p1; p += IN_U * rows) {
                    u32x4 xr[IN_U];
                    float gv[IN_U][BWD ? VEC : 1];
#pragma unroll
                    for (int u = 0; u < IN_U; ++u) xr[u] = ld16(x + base + (size_t)(p + u * rows) * pstep);
                    if (BWD) {
#pragma unroll
                        for (int u = 0; u < IN_U; ++u) in_load_grad<T, TG, (IN_NT & 4) != 0>(dy, (base + (size_t)(p + u * rows) * pstep) / sizeof(T), gv[u]);
                    }
#pragma unroll
                    for (int u = 0; u < IN_U; ++u) accumulate(xr[u], gv[u]);
                }
                for (; p < p1; p += rows) {
                    const size_t off = base + (size_t)p * pstep;
                    float gv[BWD ? VEC : 1];
                    const u32x4 xr = ld16(x + off);
                    if (BWD) in_load_grad<T, TG>(dy, off / sizeof(T), gv);
                    accumulate(xr, gv);
                }
            });
#pragma unroll
        for (int e = 0; e < VEC; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][VEC + e] = s2[e]; }
        __syncthreads();
        // fixed-order combine over the pixel rows of each channel vector
        for (int item = threadIdx.x; item < lanes * VEC; item += 256) {
            int l = item / VEC, e = item % VEC;
            AccT a = (AccT)0, b = (AccT)0;
            for (int r = 0; r < rows; ++r) { a += red[r * lanes + l][e]; b += red[r * lanes + l][VEC + e]; }
            int c = (cvb + l) * VEC + e;
            size_t o = (((size_t)n * chunks + chunk) * C + c) * 2;
            partial[o] = (float)a; partial[o + 1] = (float)b;
        }
        __syncthreads();
    }
}

// Finalize: one 1024-thread block per (32 channels, image); 32 chunk-lanes per channel, each with up to four independent
// loads in flight, combined in fixed order (deterministic).  (8 lanes walking 16 chunks one dependent L2 load after the
// other took 5-6 us per launch -- 250 launches per cycle step, 3.7 % of it.)
#define FIN_CH 32
#define FIN_LANES 32
__device__ inline void fin_reduce(const float* partial, int n, int c, int chunks, int C, int lane, double& s1, double& s2) {
    s1 = 0.0; s2 = 0.0;
    for (int k0 = lane; k0 < chunks; k0 += FIN_LANES * 4) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * FIN_LANES;
            v[u] = k < chunks ? *reinterpret_cast<const float2*>(partial + (((size_t)n * chunks + k) * C + c) * 2) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s1 += (double)v[u].x; s2 += (double)v[u].y; }
    }
}

__global__ __launch_bounds__(1024) void in_finalize_fwd_kernel(const float* partial, float* stats, int64_t HW, int C, int chunks, float eps) {
    __shared__ double red[FIN_LANES][FIN_CH][2];
    const int tx = threadIdx.x % FIN_CH, ty = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + tx, n = blockIdx.y;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) fin_reduce(partial, n, c, chunks, C, ty, s1, s2);
    red[ty][tx][0] = s1; red[ty][tx][1] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        for (int l = 1; l < FIN_LANES; ++l) { s1 += red[l][tx][0]; s2 += red[l][tx][1]; }
        double mean = s1 / (double)HW, var = s2 / (double)HW - mean * mean;
        if (var < 0.0) var = 0.0;
        size_t i = (size_t)n * C + c;
        stats[i * 2] = (float)mean;
        stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// sums[n][c] = (sum g, sum g*xhat) / HW ; tot[n][c] = the raw sums (for dgamma/dbeta)
__global__ __launch_bounds__(1024) void in_finalize_bwd_kernel(const float* partial, float* sums, float* tot, int64_t HW, int C, int chunks) {
    __shared__ double red[FIN_LANES][FIN_CH][2];
    const int tx = threadIdx.x % FIN_CH, ty = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + tx, n = blockIdx.y;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) fin_reduce(partial, n, c, chunks, C, ty, s1, s2);
    red[ty][tx][0] = s1; red[ty][tx][1] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        for (int l = 1; l < FIN_LANES; ++l) { s1 += red[l][tx][0]; s2 += red[l][tx][1]; }
        size_t i = ((size_t)n * C + c) * 2;
        sums[i] = (float)(s1 / (double)HW); sums[i + 1] = (float)(s2 / (double)HW);
        tot[i] = (float)s1; tot[i + 1] = (float)s2;
    }
}

// dgamma[c] (+)= sum_n sum g*xhat ; dbeta[c] (+)= sum_n sum g   (fixed order over n -> deterministic).  Run by the
// first block of the backward apply kernel (a separate launch cost ~5 us per instance norm for a few hundred FLOPs).
struct InParamGrad { const float* tot; float* dgamma; float* dbeta; int N, Cr, accumulate; };
__device__ inline void in_param_grad(const InParamGrad& g, int C, const InSplit& sp) {
    const int n1 = g.N < sp.nsplit ? g.N : sp.nsplit;
    for (int c = threadIdx.x; c < g.Cr; c += blockDim.x) {
        double tg = 0.0, tb = 0.0;
        for (int n = 0; n < n1; ++n) { tb += (double)g.tot[((size_t)n * C + c) * 2]; tg += (double)g.tot[((size_t)n * C + c) * 2 + 1]; }
        g.dgamma[c] = g.accumulate ? g.dgamma[c] + (float)tg : (float)tg;
        g.dbeta[c] = g.accumulate ? g.dbeta[c] + (float)tb : (float)tb;
        if (n1 < g.N) {                                  // the second network's images
            tg = 0.0; tb = 0.0;
            for (int n = n1; n < g.N; ++n) { tb += (double)g.tot[((size_t)n * C + c) * 2]; tg += (double)g.tot[((size_t)n * C + c) * 2 + 1]; }
            sp.dgamma2[c] = g.accumulate ? sp.dgamma2[c] + (float)tg : (float)tg;
            sp.dbeta2[c] = g.accumulate ? sp.dbeta2[c] + (float)tb : (float)tb;
        }
    }
}

template <typename T, bool BWD, typename TG = T>
__global__ __launch_bounds__(256) void in_apply_kernel(const char* x, const char* dy, const char* residual, const float* gamma,
                                                       const float* beta, const float* stats, const float* sums, char* out,
                                                       int64_t HW, int C, int rows_per_block, int act, float leak, InParamGrad pg, InSplit sp) {
    constexpr int VEC = ET<T>::VEC;
    const int CV = C / VEC;
    const int n = blockIdx.y;
    if (BWD && blockIdx.x == 0 && blockIdx.y == 0) in_param_grad(pg, C, sp);
    if (n >= sp.nsplit) { gamma = sp.gamma2; beta = sp.beta2; }
    const int64_t p0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t p1 = p0 + rows_per_block < HW ? p0 + rows_per_block : HW;
    for (int cvb = 0; cvb < CV; cvb += 256) {
        const int lanes = CV - cvb < 256 ? CV - cvb : 256;
        const int rows = 256 / lanes;
        const int cv = cvb + (int)(threadIdx.x % lanes), prow = threadIdx.x / lanes;
        if (prow >= rows) continue;
        float A[VEC], B[VEC], mu[VEC], rs[VEC], gm[VEC], bt[VEC], m1[VEC], m2[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            int c = cv * VEC + e;
            mu[e] = stats[((size_t)n * C + c) * 2]; rs[e] = stats[((size_t)n * C + c) * 2 + 1];
            gm[e] = gamma[c]; bt[e] = beta[c];
            A[e] = gm[e] * rs[e]; B[e] = bt[e] - mu[e] * A[e];
            if (BWD) { m1[e] = sums[((size_t)n * C + c) * 2]; m2[e] = sums[((size_t)n * C + c) * 2 + 1]; }
        }
        act_dispatch(act, [&](auto act_c) {
            constexpr int ACT = decltype(act_c)::value;
            auto sweep = [&](auto res_c) {
                constexpr bool RES = decltype(res_c)::value;
                auto one = [&](const u32x4& xr, const u32x4& rr, const float* gv, size_t off) {
                    float xv[VEC], o[VEC];
                    ET<T>::unpack(xr, xv);
                    if (!BWD) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) o[e] = act_apply_c<ACT>(xv[e] * A[e] + B[e], leak);
                        if (RES) {
                            float rv[VEC];
                            ET<T>::unpack(rr, rv);
#pragma unroll
                            for (int e = 0; e < VEC; ++e) o[e] += rv[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            float xh = (xv[e] - mu[e]) * rs[e];
                            float g = gv[e] * act_grad_c<ACT>(gm[e] * xh + bt[e], leak);
                            o[e] = A[e] * (g - m1[e] - xh * m2[e]);
                        }
                    }
                    st16(out + off, ET<T>::pack(o));
                };
                const size_t base = ((size_t)n * HW * C + (size_t)cv * VEC) * sizeof(T), pstep = (size_t)C * sizeof(T);
                int64_t p = p0 + prow;
                for (; p + (IN_U - 1) * rows < p1; p += IN_U * rows) {      // IN_U pixels per trip, all loads first (see in_partial_kernel)
                    u32x4 xr[IN_U], rr[IN_U];
                    float gv[IN_U][BWD ? VEC : 1];
#pragma unroll
                    for (int u = 0; u < IN_U; ++u) {
                        const size_t off = base + (size_t)(p + u * rows) * pstep;
                        xr[u] = (IN_NT & (BWD ? 2 : 1)) ? ld16_nt(x + off) : ld16(x + off);
                        if (RES) rr[u] = (IN_NT & 8) ? ld16_nt(residual + off) : ld16(residual + off);
                    }
                    if (BWD) {
#pragma unroll
                        for (int u = 0; u < IN_U; ++u) in_load_grad<T, TG>(dy, (base + (size_t)(p + u * rows) * pstep) / sizeof(T), gv[u]);
                    }
#pragma unroll
                    for (int u = 0; u < IN_U; ++u) one(xr[u], rr[u], gv[u], base + (size_t)(p + u * rows) * pstep);
                }
                for (; p < p1; p += rows) {
                    const size_t off = base + (size_t)p * pstep;
                    float gv[BWD ? VEC : 1];
                    const u32x4 xr = (IN_NT & (BWD ? 2 : 1)) ? ld16_nt(x + off) : ld16(x + off);
                    u32x4 rr = xr;
                    if (RES) rr = (IN_NT & 8) ? ld16_nt(residual + off) : ld16(residual + off);
                    if (BWD) in_load_grad<T, TG, (IN_NT & 4) != 0>(dy, off / sizeof(T), gv);
                    one(xr, rr, gv, off);
                }
            };
            if (!BWD && residual) sweep(std::true_type{}); else sweep(std::false_type{});
        });
    }
}


// -------------------------------------------------------------------------------------------------
// Small maps (HW <= IN_FUSED_MAXHW = 512 pixels: the discriminator tail, module.py:296-309; at 2048 pixels the serial
// walk of a block over its slab is already slower than the split kernels): one block owns a slab of IN_CVB channel
// vectors of one image and does both passes itself -- statistics, then the apply pass over the same (L2-resident)
// pixels -- in ONE launch instead of three or four; the split kernels above are launch-latency bound on these
// (20 us for a 0.5 MB tensor).  Same arithmetic order inside a block every run -> deterministic.
// -------------------------------------------------------------------------------------------------
#define IN_CVB 4
#define IN_FUSED_MAXHW 512

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void in_fused_small_kernel(const char* x, const char* dy, const char* residual, const float* gamma,
                                                             const float* beta, float* stats, float* tot, char* out,
                                                             int64_t HW, int C, float eps, int act, float leak, InSplit sp) {
    constexpr int VEC = ET<T>::VEC;
    constexpr int ROWS = 256 / IN_CVB;
    const int CV = C / VEC;
    const int n = blockIdx.y;
    if (n >= sp.nsplit) { gamma = sp.gamma2; beta = sp.beta2; }
    const int cvl = threadIdx.x % IN_CVB, prow = threadIdx.x / IN_CVB;
    const int cv = blockIdx.x * IN_CVB + cvl;
    const bool live = cv < CV;
    using AccT = InAcc<T>;
    __shared__ AccT red[256][2 * VEC + 1];
    __shared__ float sh_a[IN_CVB * VEC], sh_b[IN_CVB * VEC];

    float mu[VEC], rs[VEC], gm[VEC], bt[VEC];
    if (live) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int c = cv * VEC + e;
            gm[e] = gamma[c]; bt[e] = beta[c];
            if (BWD) { mu[e] = stats[((size_t)n * C + c) * 2]; rs[e] = stats[((size_t)n * C + c) * 2 + 1]; }
        }
    }
    AccT s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s1[e] = s2[e] = (AccT)0;
    if (live)
        for (int64_t p = prow; p < HW; p += ROWS) {
            const size_t off = (((size_t)n * HW + p) * C + (size_t)cv * VEC) * sizeof(T);
            float xv[VEC];
            ET<T>::unpack(ld16(x + off), xv);
            if (!BWD) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) { s1[e] += (AccT)xv[e]; s2[e] += (AccT)xv[e] * (AccT)xv[e]; }
            } else {
                float gv[VEC];
                ET<T>::unpack(ld16(dy + off), gv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float xh = (xv[e] - mu[e]) * rs[e];
                    const float g = gv[e] * (act == SGG_ACT_RELU ? (gm[e] * xh + bt[e] > 0.f ? 1.f : 0.f) : act == SGG_ACT_LRELU ? (gm[e] * xh + bt[e] > 0.f ? 1.f : leak) : 1.f);
                    s1[e] += (AccT)g; s2[e] += (AccT)g * (AccT)xh;
                }
            }
        }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][VEC + e] = s2[e]; }
    __syncthreads();
    if (threadIdx.x < IN_CVB * VEC) {                    // fixed-order combine over the pixel rows, in f64
        const int l = threadIdx.x / VEC, e = threadIdx.x % VEC;
        double a = 0.0, b = 0.0;
        for (int r = 0; r < ROWS; ++r) { a += (double)red[r * IN_CVB + l][e]; b += (double)red[r * IN_CVB + l][VEC + e]; }
        const int c = (blockIdx.x * IN_CVB + l) * VEC + e;
        if (c < C) {
            const size_t i = ((size_t)n * C + c) * 2;
            if (!BWD) {
                const double mean = a / (double)HW;
                double var = b / (double)HW - mean * mean;
                if (var < 0.0) var = 0.0;
                const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)eps));
                stats[i] = mf; stats[i + 1] = rf;
                sh_a[threadIdx.x] = mf; sh_b[threadIdx.x] = rf;
            } else {
                tot[i] = (float)a; tot[i + 1] = (float)b;
                sh_a[threadIdx.x] = (float)(a / (double)HW); sh_b[threadIdx.x] = (float)(b / (double)HW);
            }
        }
    }
    __syncthreads();
    if (!live) return;
    float A[VEC], B[VEC], m1[VEC], m2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        if (!BWD) { mu[e] = sh_a[cvl * VEC + e]; rs[e] = sh_b[cvl * VEC + e]; }
        else { m1[e] = sh_a[cvl * VEC + e]; m2[e] = sh_b[cvl * VEC + e]; }
        A[e] = gm[e] * rs[e]; B[e] = bt[e] - mu[e] * A[e];
    }
    act_dispatch(act, [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
        for (int64_t p = prow; p < HW; p += ROWS) {
            const size_t off = (((size_t)n * HW + p) * C + (size_t)cv * VEC) * sizeof(T);
            float xv[VEC], o[VEC];
            ET<T>::unpack(ld16(x + off), xv);
            if (!BWD) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = act_apply_c<ACT>(xv[e] * A[e] + B[e], leak);
                if (residual) {
                    float rv[VEC];
                    ET<T>::unpack(ld16(residual + off), rv);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o[e] += rv[e];
                }
            } else {
                float gv[VEC];
                ET<T>::unpack(ld16(dy + off), gv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float xh = (xv[e] - mu[e]) * rs[e];
                    const float g = gv[e] * act_grad_c<ACT>(gm[e] * xh + bt[e], leak);
                    o[e] = A[e] * (g - m1[e] - xh * m2[e]);
                }
            }
            st16(out + off, ET<T>::pack(o));
        }
    });
}

__global__ void in_param_grad_kernel(InParamGrad g, int C, InSplit sp) { in_param_grad(g, C, sp); }

static bool in_use_fused(int64_t HW) {
    const int mx = sgg_config().in_fused_maxhw;
    return HW <= (mx >= 0 ? mx : IN_FUSED_MAXHW);
}

static int in_rows_per_block(int N, int64_t HW, int C, int vec) {
    // aim for >= 2048 blocks of >= 64 pixels
    int64_t r = HW * N / 2048;
    if (r < 64) r = 64;
    if (r > 4096) r = 4096;
    return (int)r;
}

extern "C" {

size_t sgg_instnorm_workspace(int N, int64_t HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    return ((size_t)N * in_chunks(HW) * C * 2 + (size_t)N * C * 4) * sizeof(float);
}

static int instnorm_fwd_impl(const void* x, const float* gamma, const float* beta, const void* residual, void* y, float* stats,
                     int N, int64_t HW, int C, float eps, int act, float leak, int dtype, void* ws, size_t ws_bytes, void* stream, InSplit sp) {
    if (!x || !gamma || !beta || !y || !stats || N <= 0 || HW <= 0 || C <= 0 || C % SGG_CPAD) return SGG_EINVAL;
    if (act == SGG_ACT_TANH) return SGG_EUNSUPPORTED;
    if (!ws || ws_bytes < sgg_instnorm_workspace(N, HW, C)) return SGG_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (dtype != SGG_BF16 && dtype != SGG_F32) return SGG_EINVAL;
    if (in_use_fused(HW)) {
        const int vec = dtype == SGG_BF16 ? 8 : 4;
        dim3 gf((unsigned)((C / vec + IN_CVB - 1) / IN_CVB), N);
        if (dtype == SGG_BF16) hipLaunchKernelGGL((in_fused_small_kernel<bf16, false>), gf, dim3(256), 0, s, (const char*)x, nullptr, (const char*)residual, gamma, beta, stats, nullptr, (char*)y, HW, C, eps, act, leak, sp);
        else hipLaunchKernelGGL((in_fused_small_kernel<float, false>), gf, dim3(256), 0, s, (const char*)x, nullptr, (const char*)residual, gamma, beta, stats, nullptr, (char*)y, HW, C, eps, act, leak, sp);
        return sgg_check_launch();
    }
    int chunks = in_chunks(HW);
    float* partial = (float*)ws;
    int rpb = in_rows_per_block(N, HW, C, 0);
    dim3 gp(chunks, N), ga((unsigned)((HW + rpb - 1) / rpb), N);
    if (dtype == SGG_BF16) {
        hipLaunchKernelGGL((in_partial_kernel<bf16, false>), gp, dim3(256), 0, s, (const char*)x, nullptr, gamma, beta, nullptr, partial, HW, C, chunks, in_rows_per_chunk(HW), act, leak, sp);
        hipLaunchKernelGGL(in_finalize_fwd_kernel, dim3((C + FIN_CH - 1) / FIN_CH, N), dim3(1024), 0, s, partial, stats, HW, C, chunks, eps);
        sgg_launch_timed(in_apply_kernel<bf16, false>, ga, dim3(256), 0u, s, (const char*)x, (const char*)nullptr, (const char*)residual, gamma, beta, (const float*)stats, (const float*)nullptr, (char*)y, HW, C, rpb, act, leak, InParamGrad{}, sp);
    } else if (dtype == SGG_F32) {
        hipLaunchKernelGGL((in_partial_kernel<float, false>), gp, dim3(256), 0, s, (const char*)x, nullptr, gamma, beta, nullptr, partial, HW, C, chunks, in_rows_per_chunk(HW), act, leak, sp);
        hipLaunchKernelGGL(in_finalize_fwd_kernel, dim3((C + FIN_CH - 1) / FIN_CH, N), dim3(1024), 0, s, partial, stats, HW, C, chunks, eps);
        hipLaunchKernelGGL((in_apply_kernel<float, false>), ga, dim3(256), 0, s, (const char*)x, nullptr, (const char*)residual, gamma, beta, stats, nullptr, (char*)y, HW, C, rpb, act, leak, InParamGrad{}, sp);
    } else return SGG_EINVAL;
    return sgg_check_launch();
}

static int instnorm_fwd_partial_impl(const void* x, const float* gamma, const float* beta, const void* residual, void* y, float* stats,
                             const float* partial, int chunks, int N, int64_t HW, int C, float eps, int act, float leak, int dtype,
                             void* stream, InSplit sp) {
    if (!x || !gamma || !beta || !y || !stats || !partial || chunks <= 0 || N <= 0 || HW <= 0 || C <= 0 || C % SGG_CPAD) return SGG_EINVAL;
    if (act == SGG_ACT_TANH) return SGG_EUNSUPPORTED;
    if (dtype != SGG_BF16 && dtype != SGG_F32) return SGG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int rpb = in_rows_per_block(N, HW, C, 0);
    dim3 ga((unsigned)((HW + rpb - 1) / rpb), N);
    hipLaunchKernelGGL(in_finalize_fwd_kernel, dim3((C + FIN_CH - 1) / FIN_CH, N), dim3(1024), 0, s, partial, stats, HW, C, chunks, eps);
    if (dtype == SGG_BF16) sgg_launch_timed(in_apply_kernel<bf16, false>, ga, dim3(256), 0u, s, (const char*)x, (const char*)nullptr, (const char*)residual, gamma, beta, (const float*)stats, (const float*)nullptr, (char*)y, HW, C, rpb, act, leak, InParamGrad{}, sp);
    else hipLaunchKernelGGL((in_apply_kernel<float, false>), ga, dim3(256), 0, s, (const char*)x, nullptr, (const char*)residual, gamma, beta, stats, nullptr, (char*)y, HW, C, rpb, act, leak, InParamGrad{}, sp);
    return sgg_check_launch();
}

static int instnorm_bwd_impl(const void* dy, const void* x, const float* gamma, const float* beta, const float* stats, void* dx,
                     float* dgamma, float* dbeta, int N, int64_t HW, int C, int C_real, int accumulate, int act, float leak,
                     int dtype, void* ws, size_t ws_bytes, void* stream, InSplit sp) {
    if (!dy || !x || !gamma || !beta || !stats || !dx || !dgamma || !dbeta || N <= 0 || HW <= 0 || C <= 0 || C % SGG_CPAD || C_real <= 0 || C_real > C) return SGG_EINVAL;
    if (act == SGG_ACT_TANH) return SGG_EUNSUPPORTED;
    if (!ws || ws_bytes < sgg_instnorm_workspace(N, HW, C)) return SGG_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int chunks = in_chunks(HW);
    float* partial = (float*)ws;
    float* sums = partial + (size_t)N * chunks * C * 2;
    float* tot = sums + (size_t)N * C * 2;
    int rpb = in_rows_per_block(N, HW, C, 0);
    dim3 gp(chunks, N), ga((unsigned)((HW + rpb - 1) / rpb), N);
    const InParamGrad pg{tot, dgamma, dbeta, N, C_real, accumulate};
    if (dtype != SGG_BF16 && dtype != SGG_F32) return SGG_EINVAL;
    if (in_use_fused(HW)) {
        const int vec = dtype == SGG_BF16 ? 8 : 4;
        dim3 gf((unsigned)((C / vec + IN_CVB - 1) / IN_CVB), N);
        if (dtype == SGG_BF16) hipLaunchKernelGGL((in_fused_small_kernel<bf16, true>), gf, dim3(256), 0, s, (const char*)x, (const char*)dy, nullptr, gamma, beta, (float*)stats, tot, (char*)dx, HW, C, 0.f, act, leak, sp);
        else hipLaunchKernelGGL((in_fused_small_kernel<float, true>), gf, dim3(256), 0, s, (const char*)x, (const char*)dy, nullptr, gamma, beta, (float*)stats, tot, (char*)dx, HW, C, 0.f, act, leak, sp);
        hipLaunchKernelGGL(in_param_grad_kernel, dim3(1), dim3(256), 0, s, pg, C, sp);
        return sgg_check_launch();
    }
    if (dtype == SGG_BF16) {
        hipLaunchKernelGGL((in_partial_kernel<bf16, true>), gp, dim3(256), 0, s, (const char*)x, (const char*)dy, gamma, beta, stats, partial, HW, C, chunks, in_rows_per_chunk(HW), act, leak, sp);
        hipLaunchKernelGGL(in_finalize_bwd_kernel, dim3((C + FIN_CH - 1) / FIN_CH, N), dim3(1024), 0, s, partial, sums, tot, HW, C, chunks);
        hipLaunchKernelGGL((in_apply_kernel<bf16, true>), ga, dim3(256), 0, s, (const char*)x, (const char*)dy, nullptr, gamma, beta, stats, sums, (char*)dx, HW, C, rpb, act, leak, pg, sp);
    } else if (dtype == SGG_F32) {
        hipLaunchKernelGGL((in_partial_kernel<float, true>), gp, dim3(256), 0, s, (const char*)x, (const char*)dy, gamma, beta, stats, partial, HW, C, chunks, in_rows_per_chunk(HW), act, leak, sp);
        hipLaunchKernelGGL(in_finalize_bwd_kernel, dim3((C + FIN_CH - 1) / FIN_CH, N), dim3(1024), 0, s, partial, sums, tot, HW, C, chunks);
        hipLaunchKernelGGL((in_apply_kernel<float, true>), ga, dim3(256), 0, s, (const char*)x, (const char*)dy, nullptr, gamma, beta, stats, sums, (char*)dx, HW, C, rpb, act, leak, pg, sp);
    } else return SGG_EINVAL;
    return sgg_check_launch();
}

// Mixed mode: dy is f32, x / dx are bf16 (the gradient chain between the instance norms of the residual blocks stays f32:
// the norm backward subtracts the mean and the xhat-correlated part of dy, so a bf16 rounding of dy -- relative to dy, not
// to what is left of it -- is amplified there layer after layer).  Same passes as sgg_instnorm_bwd.
static int instnorm_bwd_mixed_impl(const float* dy, const void* x, const float* gamma, const float* beta, const float* stats, void* dx,
                           float* dgamma, float* dbeta, int N, int64_t HW, int C, int C_real, int accumulate, int act, float leak,
                           void* ws, size_t ws_bytes, void* stream, InSplit sp) {
    if (!dy || !x || !gamma || !beta || !stats || !dx || !dgamma || !dbeta || N <= 0 || HW <= 0 || C <= 0 || C % SGG_CPAD || C_real <= 0 || C_real > C) return SGG_EINVAL;
    if (act == SGG_ACT_TANH) return SGG_EUNSUPPORTED;
    if (!ws || ws_bytes < sgg_instnorm_workspace(N, HW, C)) return SGG_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int chunks = in_chunks(HW);
    float* partial = (float*)ws;
    float* sums = partial + (size_t)N * chunks * C * 2;
    float* tot = sums + (size_t)N * C * 2;
    int rpb = in_rows_per_block(N, HW, C, 0);
    dim3 gp(chunks, N), ga((unsigned)((HW + rpb - 1) / rpb), N);
    const InParamGrad pg{tot, dgamma, dbeta, N, C_real, accumulate};
    hipLaunchKernelGGL((in_partial_kernel<bf16, true, float>), gp, dim3(256), 0, s, (const char*)x, (const char*)dy, gamma, beta, stats, partial, HW, C, chunks, in_rows_per_chunk(HW), act, leak, sp);
    hipLaunchKernelGGL(in_finalize_bwd_kernel, dim3((C + FIN_CH - 1) / FIN_CH, N), dim3(1024), 0, s, partial, sums, tot, HW, C, chunks);
    hipLaunchKernelGGL((in_apply_kernel<bf16, true, float>), ga, dim3(256), 0, s, (const char*)x, (const char*)dy, nullptr, gamma, beta, stats, sums, (char*)dx, HW, C, rpb, act, leak, pg, sp);
    return sgg_check_launch();
}

static int instnorm_bwd_partial_impl(const void* dy, const void* x, const float* gamma, const float* beta, const float* stats, void* dx,
                             float* dgamma, float* dbeta, const float* partial, int chunks, int N, int64_t HW, int C, int C_real,
                             int accumulate, int act, float leak, int dtype, void* ws, size_t ws_bytes, void* stream, InSplit sp) {
    if (!dy || !x || !gamma || !beta || !stats || !dx || !dgamma || !dbeta || !partial || chunks <= 0 || N <= 0 || HW <= 0 || C <= 0 ||
        C % SGG_CPAD || C_real <= 0 || C_real > C) return SGG_EINVAL;
    if (act == SGG_ACT_TANH) return SGG_EUNSUPPORTED;
    if (dtype != SGG_BF16 && dtype != SGG_F32) return SGG_EINVAL;
    if (!ws || ws_bytes < (size_t)N * C * 4 * sizeof(float)) return SGG_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* sums = (float*)ws;
    float* tot = sums + (size_t)N * C * 2;
    int rpb = in_rows_per_block(N, HW, C, 0);
    dim3 ga((unsigned)((HW + rpb - 1) / rpb), N);
    const InParamGrad pg{tot, dgamma, dbeta, N, C_real, accumulate};
    hipLaunchKernelGGL(in_finalize_bwd_kernel, dim3((C + FIN_CH - 1) / FIN_CH, N), dim3(1024), 0, s, partial, sums, tot, HW, C, chunks);
    if (dtype == SGG_BF16) hipLaunchKernelGGL((in_apply_kernel<bf16, true>), ga, dim3(256), 0, s, (const char*)x, (const char*)dy, nullptr, gamma, beta, stats, sums, (char*)dx, HW, C, rpb, act, leak, pg, sp);
    else hipLaunchKernelGGL((in_apply_kernel<float, true>), ga, dim3(256), 0, s, (const char*)x, (const char*)dy, nullptr, gamma, beta, stats, sums, (char*)dx, HW, C, rpb, act, leak, pg, sp);
    return sgg_check_launch();
}

// ---- public entry points: one network, or two networks of the same shape stacked on the batch dimension ("pair": images
// 0..nsplit-1 use gamma/beta and add to dgamma/dbeta, the rest use gamma2/beta2 and dgamma2/dbeta2)
int sgg_instnorm_fwd(const void* x, const float* gamma, const float* beta, const void* residual, void* y, float* stats,
                     int N, int64_t HW, int C, float eps, int act, float leak, int dtype, void* ws, size_t ws_bytes, void* stream) {
    return instnorm_fwd_impl(x, gamma, beta, residual, y, stats, N, HW, C, eps, act, leak, dtype, ws, ws_bytes, stream, in_nosplit());
}
int sgg_instnorm_fwd_partial(const void* x, const float* gamma, const float* beta, const void* residual, void* y, float* stats,
                             const float* partial, int chunks, int N, int64_t HW, int C, float eps, int act, float leak, int dtype,
                             void* stream) {
    return instnorm_fwd_partial_impl(x, gamma, beta, residual, y, stats, partial, chunks, N, HW, C, eps, act, leak, dtype, stream, in_nosplit());
}
int sgg_instnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* stats, void* dx,
                     float* dgamma, float* dbeta, int N, int64_t HW, int C, int C_real, int accumulate, int act, float leak,
                     int dtype, void* ws, size_t ws_bytes, void* stream) {
    return instnorm_bwd_impl(dy, x, gamma, beta, stats, dx, dgamma, dbeta, N, HW, C, C_real, accumulate, act, leak, dtype, ws, ws_bytes, stream, in_nosplit());
}
int sgg_instnorm_bwd_mixed(const float* dy, const void* x, const float* gamma, const float* beta, const float* stats, void* dx,
                           float* dgamma, float* dbeta, int N, int64_t HW, int C, int C_real, int accumulate, int act, float leak,
                           void* ws, size_t ws_bytes, void* stream) {
    return instnorm_bwd_mixed_impl(dy, x, gamma, beta, stats, dx, dgamma, dbeta, N, HW, C, C_real, accumulate, act, leak, ws, ws_bytes, stream, in_nosplit());
}
int sgg_instnorm_bwd_partial(const void* dy, const void* x, const float* gamma, const float* beta, const float* stats, void* dx,
                             float* dgamma, float* dbeta, const float* partial, int chunks, int N, int64_t HW, int C, int C_real,
                             int accumulate, int act, float leak, int dtype, void* ws, size_t ws_bytes, void* stream) {
    return instnorm_bwd_partial_impl(dy, x, gamma, beta, stats, dx, dgamma, dbeta, partial, chunks, N, HW, C, C_real, accumulate, act, leak, dtype, ws, ws_bytes, stream, in_nosplit());
}
// (mean, rstd)[N][C] from per-chunk (sum, sumsq) rows -- the finalize step of sgg_instnorm_fwd_partial on its own, for a
// consumer that applies the norm itself (sgg_conv2d_fwd_stats_normload)
int sgg_instnorm_finalize(const float* partial, int chunks, float* stats, int N, int64_t HW, int C, float eps, void* stream) {
    if (!partial || !stats || chunks <= 0 || N <= 0 || HW <= 0 || C <= 0 || C % SGG_CPAD) return SGG_EINVAL;
    hipLaunchKernelGGL(in_finalize_fwd_kernel, dim3((C + FIN_CH - 1) / FIN_CH, N), dim3(1024), 0, (hipStream_t)stream, partial, stats, HW, C, chunks, eps);
    return sgg_check_launch();
}
static inline int in_pair_ok(const float* g2, const float* b2, int nsplit, int N) { return g2 && b2 && nsplit > 0 && nsplit < N; }
int sgg_instnorm_fwd_pair(const void* x, const float* gamma, const float* beta, const float* gamma2, const float* beta2, int nsplit,
                          const void* residual, void* y, float* stats, int N, int64_t HW, int C, float eps, int act, float leak,
                          int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!in_pair_ok(gamma2, beta2, nsplit, N)) return SGG_EINVAL;
    InSplit sp = in_nosplit(); sp.gamma2 = gamma2; sp.beta2 = beta2; sp.nsplit = nsplit;
    return instnorm_fwd_impl(x, gamma, beta, residual, y, stats, N, HW, C, eps, act, leak, dtype, ws, ws_bytes, stream, sp);
}
int sgg_instnorm_fwd_partial_pair(const void* x, const float* gamma, const float* beta, const float* gamma2, const float* beta2, int nsplit,
                                  const void* residual, void* y, float* stats, const float* partial, int chunks, int N, int64_t HW, int C,
                                  float eps, int act, float leak, int dtype, void* stream) {
    if (!in_pair_ok(gamma2, beta2, nsplit, N)) return SGG_EINVAL;
    InSplit sp = in_nosplit(); sp.gamma2 = gamma2; sp.beta2 = beta2; sp.nsplit = nsplit;
    return instnorm_fwd_partial_impl(x, gamma, beta, residual, y, stats, partial, chunks, N, HW, C, eps, act, leak, dtype, stream, sp);
}
int sgg_instnorm_bwd_pair(const void* dy, const void* x, const float* gamma, const float* beta, const float* gamma2, const float* beta2,
                          int nsplit, const float* stats, void* dx, float* dgamma, float* dbeta, float* dgamma2, float* dbeta2,
                          int N, int64_t HW, int C, int C_real, int accumulate, int act, float leak, int dtype,
                          void* ws, size_t ws_bytes, void* stream) {
    if (!in_pair_ok(gamma2, beta2, nsplit, N) || !dgamma2 || !dbeta2) return SGG_EINVAL;
    InSplit sp; sp.gamma2 = gamma2; sp.beta2 = beta2; sp.dgamma2 = dgamma2; sp.dbeta2 = dbeta2; sp.nsplit = nsplit;
    return instnorm_bwd_impl(dy, x, gamma, beta, stats, dx, dgamma, dbeta, N, HW, C, C_real, accumulate, act, leak, dtype, ws, ws_bytes, stream, sp);
}

}  // extern "C"
