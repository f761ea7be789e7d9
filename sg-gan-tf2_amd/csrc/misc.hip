// misc.hip -- the small HBM-bound kernels of the train step: activations, bias gradient, semantic mask
// multiply-reduce (module.py:312-314), losses (model.py:149-166), Keras-form Adam (model.py:199-207), the
// colour->class-index map (segment_class.py:60-99), one-hot + resample of the mask (utils.py:158-165,197-199)
// and the channel pad/unpad at the boundary.
#include "common.h"

// ---------------------------------------------------------------- activations / add
template <typename T, int OP>   // OP 0: act fwd (a=x)   1: act bwd (a=dy, b=y)   2: add (a+b)
__global__ __launch_bounds__(256) void eltwise_kernel(const char* a, const char* b, char* o, int64_t nvec, int act, float leak) {
    constexpr int VEC = ET<T>::VEC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float av[VEC], bv[VEC], ov[VEC];
        ET<T>::unpack(ld16(a + i * 16), av);
        if (OP != 0) ET<T>::unpack(ld16(b + i * 16), bv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (OP == 0) ov[e] = act_apply(av[e], act, leak);
            else if (OP == 1) {
                float d = 1.f;
                if (act == SGG_ACT_RELU) d = bv[e] > 0.f ? 1.f : 0.f;
                else if (act == SGG_ACT_LRELU) d = bv[e] > 0.f ? 1.f : leak;
                else if (act == SGG_ACT_TANH) d = 1.f - bv[e] * bv[e];
                ov[e] = av[e] * d;
            } else ov[e] = av[e] + bv[e];
        }
        st16(o + i * 16, ET<T>::pack(ov));
    }
}

template <int OP>
static int launch_eltwise(const void* a, const void* b, void* o, int64_t n, int act, float leak, int dtype, void* stream) {
    if (!a || !o || n < 0 || (OP != 0 && !b)) return SGG_EINVAL;
    if (n == 0) return SGG_OK;
    int vec = dtype == SGG_BF16 ? 8 : 4;
    if (n % vec) return SGG_EINVAL;
    int64_t nvec = n / vec;
    int blocks = (int)((nvec + 255) / 256); if (blocks > 4096) blocks = 4096;
    if (dtype == SGG_BF16) hipLaunchKernelGGL((eltwise_kernel<bf16, OP>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)a, (const char*)b, (char*)o, nvec, act, leak);
    else if (dtype == SGG_F32) hipLaunchKernelGGL((eltwise_kernel<float, OP>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)a, (const char*)b, (char*)o, nvec, act, leak);
    else return SGG_EINVAL;
    return sgg_check_launch();
}

// ---------------------------------------------------------------- column sums (bias gradient)
#define BG_ROWS 1024
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const char* dy, float* partial, int64_t P, int C) {
    constexpr int VEC = ET<T>::VEC;
    const int CV = C / VEC;
    // grouped call (sgg_bias_grad_group2): blockIdx.y = 1 is the second network's tensor, right behind the first one's
    dy += (size_t)blockIdx.y * P * C * sizeof(T);
    partial += (size_t)blockIdx.y * gridDim.x * C;
    const int64_t p0 = (int64_t)blockIdx.x * BG_ROWS, p1 = p0 + BG_ROWS < P ? p0 + BG_ROWS : P;
    __shared__ float red[256][VEC + 1];
    for (int cvb = 0; cvb < CV; cvb += 256) {
        const int lanes = CV - cvb < 256 ? CV - cvb : 256, rows = 256 / lanes;
        const int cv = cvb + (int)(threadIdx.x % lanes), prow = threadIdx.x / lanes;
        float s[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] = 0.f;
        if (prow < rows)
            for (int64_t p = p0 + prow; p < p1; p += rows) {
                float v[VEC];
                ET<T>::unpack(ld16(dy + ((size_t)p * C + (size_t)cv * VEC) * sizeof(T)), v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[e] += v[e];
            }
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[threadIdx.x][e] = s[e];
        __syncthreads();
        for (int item = threadIdx.x; item < lanes * VEC; item += 256) {
            int l = item / VEC, e = item % VEC;
            float acc = 0.f;
            for (int r = 0; r < rows; ++r) acc += red[r * lanes + l][e];
            partial[(size_t)blockIdx.x * C + (cvb + l) * VEC + e] = acc;
        }
        __syncthreads();
    }
}
// one 1024-thread block per 16 channels; 64 chunk-lanes per channel, four independent loads in flight each, combined in
// fixed order (the head's bias gradient has 1 024 chunks of 8 channels: 4 lanes walking them one dependent load after
// the other took 25 us)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* partial, float* out, int C, int Cr, int chunks, int accumulate, float* out2) {
    __shared__ double red[64][16];
    if (blockIdx.y) { partial += (size_t)chunks * C; out = out2; }
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    double s = 0.0;
    if (c < C)
        for (int k0 = ty; k0 < chunks; k0 += 64 * 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int k = k0 + 64 * u; v[u] = k < chunks ? partial[(size_t)k * C + c] : 0.f; }
#pragma unroll
            for (int u = 0; u < 4; ++u) s += (double)v[u];
        }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < Cr) {
        s = 0.0;
        for (int l = 0; l < 64; ++l) s += red[l][tx];
        out[c] = accumulate ? out[c] + (float)s : (float)s;
    }
}

// ---------------------------------------------------------------- semantic mask multiply-reduce
template <typename T>
__global__ void mask_reduce_fwd_kernel(const T* h4, const float* mask, float* out, int N, int hh, int hw, int mh, int mw, int Cr, int Cp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int total = N * mh * mw;
    if (i >= total) return;
    int n = i / (mh * mw), rem = i % (mh * mw), y = rem / mw, x = rem % mw;
    int hy = hh == 1 ? 0 : y, hx = hw == 1 ? 0 : x;
    const T* h = h4 + (((size_t)n * hh + hy) * hw + hx) * Cp;
    const float* m = mask + (size_t)i * Cr;
    float s = 0.f;
    for (int c = 0; c < Cr; ++c) s += (float)h[c] * m[c];
    out[i] = s;
}
template <typename T>
__global__ void mask_reduce_bwd_kernel(const float* dout, const float* mask, T* dh4, int N, int hh, int hw, int mh, int mw, int Cr, int Cp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int total = N * hh * hw * Cp;
    if (i >= total) return;
    int c = i % Cp, t = i / Cp, hx = t % hw, hy = (t / hw) % hh, n = t / (hw * hh);
    float s = 0.f;
    if (c < Cr) {
        int y0 = hh == 1 ? 0 : hy, y1 = hh == 1 ? mh : hy + 1;
        int x0 = hw == 1 ? 0 : hx, x1 = hw == 1 ? mw : hx + 1;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                size_t cell = ((size_t)n * mh + y) * mw + x;
                s += dout[cell] * mask[cell * Cr + c];
            }
    }
    dh4[i] = (T)s;
}

// ---------------------------------------------------------------- losses (single block: the maps are tiny / staged)
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* x, int64_t n, float label, float weight, float gscale,
                                                         float* loss, float* dx, int accumulate) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        float v = x[i];
        s += (double)(fmaxf(v, 0.f) - v * label + log1pf(expf(-fabsf(v))));
        float sig = 1.f / (1.f + expf(-v));
        float g = weight * gscale * (sig - label) / (float)n;
        if (dx) dx[i] = (accumulate & 2) ? dx[i] + g : g;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0 && loss) { float l = (float)(weight * red[0] / (double)n); *loss = (accumulate & 1) ? *loss + l : l; }
}

// LSGAN criterion against a constant target: mae_criterion (module.py:340-341) = mean((x - t)^2)  [sic: squared]
__global__ __launch_bounds__(256) void mse_const_kernel(const float* x, int64_t n, float target, float weight, float gscale,
                                                        float* loss, float* dx, int accumulate) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        float d = x[i] - target;
        s += (double)(d * d);
        float g = weight * gscale * 2.f * d / (float)n;
        if (dx) dx[i] = (accumulate & 2) ? dx[i] + g : g;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0 && loss) { float l = (float)(weight * red[0] / (double)n); *loss = (accumulate & 1) ? *loss + l : l; }
}

// Segmentation-edge indicator (model.py:108-119): central differences along x and y of the REFLECT-padded colour
// segmentation, |.| summed over channels, then sign -> 1 on class boundaries, 0 inside regions.  out f32 [N][H][W].
template <typename T>
__global__ __launch_bounds__(256) void seg_edge_kernel(const T* seg, float* out, int N, int H, int W, int Cr, int Cp) {
    int64_t total = (int64_t)N * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int w = (int)(i % W), h = (int)((i / W) % H), n = (int)(i / ((int64_t)W * H));
        int wl = w == 0 ? 1 : w - 1, wr = w == W - 1 ? W - 2 : w + 1;       // REFLECT (no edge repeat)
        int hu = h == 0 ? 1 : h - 1, hd = h == H - 1 ? H - 2 : h + 1;
        const T* b = seg + (size_t)n * H * W * Cp;
        float s = 0.f;
        for (int c = 0; c < Cr; ++c) {
            s += fabsf((float)b[((size_t)h * W + wr) * Cp + c] - (float)b[((size_t)h * W + wl) * Cp + c]);
            s += fabsf((float)b[((size_t)hd * W + w) * Cp + c] - (float)b[((size_t)hu * W + w) * Cp + c]);
        }
        out[i] = s > 0.f ? 1.f : 0.f;
    }
}

// Gradient-sensitive loss (module.py:325-351): Sobel gx/gy (depthwise, SAME zero padding) of `in` and `target`,
// abs_deriv = | |d(in)| - |d(target)| |, mean over the 2*C derivative channels, mean over pixels of weight*abs_deriv.
template <typename T>
__device__ inline void sobel_at(const T* img, int H, int W, int Cp, int h, int w, int c, float& gx, float& gy) {
    float v[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            int hh = h + dy - 1, ww = w + dx - 1;
            v[dy][dx] = ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) ? (float)img[((size_t)hh * W + ww) * Cp + c] : 0.f;
        }
    gx = (v[0][2] - v[0][0]) + 2.f * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0]);
    gy = (v[2][0] - v[0][0]) + 2.f * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2]);
}

// pass 1: per-pixel loss terms -> block partials; coef[p][c][2] = d loss / d (derivative of `in`) (if coef != null)
template <typename T>
__global__ __launch_bounds__(256) void gradloss_fwd_kernel(const T* in, const T* tgt, const float* weight, float* coef, float* partial,
                                                           int N, int H, int W, int Cr, int Cp, float gval) {
    int64_t total = (int64_t)N * H * W;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int w = (int)(i % W), h = (int)((i / W) % H), n = (int)(i / ((int64_t)W * H));
        const T* a = in + (size_t)n * H * W * Cp;
        const T* b = tgt + (size_t)n * H * W * Cp;
        float wt = weight[i], acc = 0.f;
        for (int c = 0; c < Cr; ++c) {
            float ax, ay, bx, by;
            sobel_at(a, H, W, Cp, h, w, c, ax, ay);
            sobel_at(b, H, W, Cp, h, w, c, bx, by);
            float dx = fabsf(ax) - fabsf(bx), dy = fabsf(ay) - fabsf(by);
            acc += fabsf(dx) + fabsf(dy);
            if (coef) {
                float sx = (dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f)) * (ax > 0.f ? 1.f : (ax < 0.f ? -1.f : 0.f));
                float sy = (dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f)) * (ay > 0.f ? 1.f : (ay < 0.f ? -1.f : 0.f));
                coef[(i * Cr + c) * 2] = gval * wt * sx;
                coef[(i * Cr + c) * 2 + 1] = gval * wt * sy;
            }
        }
        s += wt * acc;
    }
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// pass 2: d in[q][c] = sum over the 3x3 neighbours p of coef[p][c][j] * K_j[q - p + 1]   (transpose of the Sobel correlation)
template <typename T>
__global__ __launch_bounds__(256) void gradloss_bwd_kernel(const float* coef, T* din, int N, int H, int W, int Cr, int Cp, int accumulate) {
    const float KX[3][3] = {{-1.f, 0.f, 1.f}, {-2.f, 0.f, 2.f}, {-1.f, 0.f, 1.f}};
    const float KY[3][3] = {{-1.f, -2.f, -1.f}, {0.f, 0.f, 0.f}, {1.f, 2.f, 1.f}};
    int64_t total = (int64_t)N * H * W * Cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % Cp);
        int64_t px = i / Cp;
        int w = (int)(px % W), h = (int)((px / W) % H), n = (int)(px / ((int64_t)W * H));
        float g = 0.f;
        if (c < Cr) {
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) {
                    // derivative at p = q - (dy-1, dx-1) read in[q] with kernel element K[dy][dx]
                    int ph = h - (dy - 1), pw = w - (dx - 1);
                    if ((unsigned)ph < (unsigned)H && (unsigned)pw < (unsigned)W) {
                        size_t o = ((((size_t)n * H + ph) * W + pw) * Cr + c) * 2;
                        g += coef[o] * KX[dy][dx] + coef[o + 1] * KY[dy][dx];
                    }
                }
        }
        din[i] = (T)(accumulate ? (float)din[i] + g : g);
    }
}

// The same two passes with one THREAD PER PIXEL and 16-byte pixel loads (the image tensors here have <= 4 real channels in a
// 16-byte vector: 8 bf16 / 4 f32): 18 vector loads per pixel instead of 54 scalar ones forward, one thread instead of Cp backward.
// Same pixel -> thread assignment and the same summation order per pixel as the scalar kernels: bit-identical results.
template <typename T>
__global__ __launch_bounds__(256) void gradloss_fwd_vec_kernel(const char* in, const char* tgt, const float* weight, float* coef, float* partial,
                                                               int N, int H, int W, int Cr, int Cp, float gval) {
    constexpr int VEC = ET<T>::VEC;
    const int64_t total = (int64_t)N * H * W;
    const size_t pstride = (size_t)Cp * sizeof(T);
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W), h = (int)((i / W) % H);
        u32x4 ra[3][3], rb[3][3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int hh = h + dy - 1, ww = w + dx - 1;
                const bool ok = (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
                const int64_t o = i + (int64_t)(dy - 1) * W + (dx - 1);             // same image: (hh, ww) is inside it when ok
                ra[dy][dx] = ok ? ld16(in + (size_t)o * pstride) : zero16();
                rb[dy][dx] = ok ? ld16(tgt + (size_t)o * pstride) : zero16();
            }
        float va[3][3][VEC], vb[3][3][VEC];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) { ET<T>::unpack(ra[dy][dx], va[dy][dx]); ET<T>::unpack(rb[dy][dx], vb[dy][dx]); }
        const float wt = weight[i];
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c >= Cr) break;
            const float ax = (va[0][2][c] - va[0][0][c]) + 2.f * (va[1][2][c] - va[1][0][c]) + (va[2][2][c] - va[2][0][c]);
            const float ay = (va[2][0][c] - va[0][0][c]) + 2.f * (va[2][1][c] - va[0][1][c]) + (va[2][2][c] - va[0][2][c]);
            const float bx = (vb[0][2][c] - vb[0][0][c]) + 2.f * (vb[1][2][c] - vb[1][0][c]) + (vb[2][2][c] - vb[2][0][c]);
            const float by = (vb[2][0][c] - vb[0][0][c]) + 2.f * (vb[2][1][c] - vb[0][1][c]) + (vb[2][2][c] - vb[0][2][c]);
            const float dx = fabsf(ax) - fabsf(bx), dy = fabsf(ay) - fabsf(by);
            acc += fabsf(dx) + fabsf(dy);
            if (coef) {
                const float sx = (dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f)) * (ax > 0.f ? 1.f : (ax < 0.f ? -1.f : 0.f));
                const float sy = (dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f)) * (ay > 0.f ? 1.f : (ay < 0.f ? -1.f : 0.f));
                *reinterpret_cast<float2*>(coef + (i * Cr + c) * 2) = make_float2(gval * wt * sx, gval * wt * sy);
            }
        }
        s += wt * acc;
    }
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

template <typename T>
__global__ __launch_bounds__(256) void gradloss_bwd_vec_kernel(const float* coef, char* din, int N, int H, int W, int Cr, int Cp, int accumulate) {
    constexpr int VEC = ET<T>::VEC;
    const float KX[3][3] = {{-1.f, 0.f, 1.f}, {-2.f, 0.f, 2.f}, {-1.f, 0.f, 1.f}};
    const float KY[3][3] = {{-1.f, -2.f, -1.f}, {0.f, 0.f, 0.f}, {1.f, 2.f, 1.f}};
    const int64_t total = (int64_t)N * H * W;
    const size_t pstride = (size_t)Cp * sizeof(T);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W), h = (int)((i / W) % H);
        float g[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) g[c] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                // derivative at p = q - (dy-1, dx-1) read in[q] with kernel element K[dy][dx]
                const int ph = h - (dy - 1), pw = w - (dx - 1);
                if ((unsigned)ph < (unsigned)H && (unsigned)pw < (unsigned)W) {
                    const float* cp = coef + (size_t)(i - (int64_t)(dy - 1) * W - (dx - 1)) * Cr * 2;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (c >= Cr) break;
                        const float2 v = *reinterpret_cast<const float2*>(cp + c * 2);
                        g[c] += v.x * KX[dy][dx] + v.y * KY[dy][dx];
                    }
                }
            }
        char* o = din + (size_t)i * pstride;
        if (accumulate) {
            float old[VEC];
            ET<T>::unpack(ld16(o), old);
#pragma unroll
            for (int c = 0; c < VEC; ++c) g[c] = (float)(T)(old[c] + g[c]);      // channels >= Cr: old + 0 (the scalar kernel's din[i] + g)
            st16(o, ET<T>::pack(g));
        } else {
            st16(o, ET<T>::pack(g));
            for (int v = 1; v * VEC < Cp; ++v) st16(o + v * 16, zero16());
        }
    }
}

#define L1_ROWS 2048
template <typename T>
__global__ __launch_bounds__(256) void l1_partial_kernel(const char* a, const char* b, char* db, float* partial, int64_t nvec,
                                                         int Cr, int Cp, float gval, int accdb) {
    constexpr int VEC = ET<T>::VEC;
    const int64_t i0 = (int64_t)blockIdx.x * L1_ROWS;
    const int64_t i1 = i0 + L1_ROWS < nvec ? i0 + L1_ROWS : nvec;
    float s = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        float av[VEC], bv[VEC], g[VEC];
        ET<T>::unpack(ld16(a + i * 16), av);
        ET<T>::unpack(ld16(b + i * 16), bv);
        int c0 = (int)((i * VEC) % Cp);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float d = av[e] - bv[e];
            bool real = (c0 + e) < Cr;
            s += real ? fabsf(d) : 0.f;
            g[e] = real ? (d > 0.f ? -gval : (d < 0.f ? gval : 0.f)) : 0.f;      // d/db |a-b| = -sign(a-b)
        }
        if (db) {
            if (accdb) { float o[VEC]; ET<T>::unpack(ld16(db + i * 16), o);
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] += o[e]; }
            st16(db + i * 16, ET<T>::pack(g));
        }
    }
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void l1_final_kernel(const float* partial, int chunks, double scale, float* loss, int accumulate) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < chunks; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { float l = (float)(red[0] * scale); *loss = accumulate ? *loss + l : l; }
}

// ---------------------------------------------------------------- Adam (Keras form), flat buffer
__global__ __launch_bounds__(256) void adam_kernel(float* th, const float* g, float* m, float* v, int64_t n, float lr_t,
                                                   float b1, float b2, float eps, float gs) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * gs;
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        th[i] = th[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}

// Same update with the step number read from a device-resident counter (Keras keeps `optimizer.iterations` as a variable
// too).  `state` is int64[2]: [0] = iterations, [1] = scratch (the bits of lr_t for the step being applied).  A 1-thread
// launch evaluates lr_t in double from t = iterations + 1 (same formula as the host path of sgg_adam) and bumps the
// counter; the update kernel reads the float.  The host never touches t, so the pair of launches can sit inside a captured
// HIP graph and be replayed.  (Evaluating pow() in every block of the update kernel cost +18 us per launch.)
__global__ void adam_prep_kernel(int64_t* state, float lr, float b1, float b2) {
    const double t = (double)(state[0] + 1);
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
    state[0] += 1;
    reinterpret_cast<float*>(state + 1)[0] = lr_t;
}
__global__ __launch_bounds__(256) void adam_iter_kernel(float* th, const float* g, float* m, float* v, int64_t n, const int64_t* state,
                                                        float b1, float b2, float eps, float gs) {
    const float lr_t = reinterpret_cast<const float*>(state + 1)[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * gs;
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        th[i] = th[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}

// ---------------------------------------------------------------- colour -> class index (integer, bit exact)
// segment_class.py:63-66: 21 colours -> {1..7}; default 0.  Keys are 24-bit (R<<16|G<<8|B).
#define SGG_SEG_KEYS 0x804080, 0xF423E8, 0xFAAAA0, 0xE6968C, 0x464646, 0x66669C, 0xBE9999, 0xB4A5B4, 0x966464, 0x96785A, \
                     0x6B8E23, 0x4682B4, 0xDC143C, 0xFF0000, 0x00008E, 0x000046, 0x003C64, 0x00005A, 0x00006E, 0x0000E6, 0x770B20
#define SGG_SEG_VALS 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 7, 6, 2, 2, 1, 1, 1, 1, 1, 3, 3
__constant__ uint32_t kSegKeys[21] = {SGG_SEG_KEYS};
__constant__ uint8_t kSegVals[21] = {SGG_SEG_VALS};
static const uint32_t hSegKeys[21] = {SGG_SEG_KEYS};     // host copy, exported for the CPU-side table check
static const uint8_t hSegVals[21] = {SGG_SEG_VALS};

__global__ __launch_bounds__(256) void seg_class_kernel(const uint8_t* rgb, int ch, int64_t n, uint8_t* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint8_t* p = rgb + i * ch;
        uint32_t key = ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2];
        uint8_t v = 0;
#pragma unroll
        for (int k = 0; k < 21; ++k) v = key == kSegKeys[k] ? kSegVals[k] : v;
        out[i] = v;
    }
}

// Four pixels per thread: 12 (RGB) or 16 (RGBA) contiguous input bytes as three or four dword loads, one dword store of the four
// class indices -- every access of a wave is a contiguous run (the byte-per-thread form above touches each 128-byte line from 32
// lanes with 3 one-byte loads per pixel).  Needs 4-byte aligned pointers; the tail (n mod 4 pixels) takes the scalar kernel.
template <int CH>
__global__ __launch_bounds__(256) void seg_class_vec4_kernel(const uint32_t* rgb, int64_t ngroups, uint32_t* out) {
    for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < ngroups; gi += (int64_t)gridDim.x * blockDim.x) {
        uint32_t w[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) w[k] = rgb[gi * CH + k];
        uint32_t packed = 0;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            uint32_t key = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int byte = px * CH + c;                       // compile-time position in the 4 * CH input bytes
                key = (key << 8) | ((w[byte >> 2] >> (8 * (byte & 3))) & 0xffu);
            }
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 21; ++k) v = key == kSegKeys[k] ? (uint32_t)kSegVals[k] : v;
            packed |= v << (8 * px);
        }
        out[gi] = packed;
    }
}

__global__ void onehot_resample_kernel(const uint8_t* idx, float* mask, int N, int H, int W, int oh, int ow, int nc) {
    int64_t total = (int64_t)N * oh * ow * nc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % nc);
        int64_t t = i / nc;
        int x = (int)(t % ow), y = (int)((t / ow) % oh), n = (int)(t / ((int64_t)ow * oh));
        // align-corners nearest, round half up; exact rational arithmetic: floor((2*y*(H-1) + (oh-1)) / (2*(oh-1)))
        int sy = oh == 1 ? 0 : (int)((2ll * y * (H - 1) + (oh - 1)) / (2ll * (oh - 1)));
        int sx = ow == 1 ? 0 : (int)((2ll * x * (W - 1) + (ow - 1)) / (2ll * (ow - 1)));
        mask[i] = idx[((size_t)n * H + sy) * W + sx] == c ? 1.f : 0.f;
    }
}

template <typename T>
__global__ void pad_channels_kernel(const float* src, T* dst, int64_t P, int Cs, int Cd) {
    int64_t total = P * Cd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % Cd);
        int64_t p = i / Cd;
        dst[i] = (T)(c < Cs ? src[p * Cs + c] : 0.f);
    }
}
template <typename T>
__global__ void unpad_channels_kernel(const T* src, float* dst, int64_t P, int Cs, int Cd) {
    int64_t total = P * Cd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % Cd);
        int64_t p = i / Cd;
        dst[i] = (float)src[p * Cs + c];
    }
}

// ---------------------------------------------------------------- evaluation (metric.py:18-47,71-77): integer, bit exact
// hist[n_class * t + p] += 1 for every pixel with 0 <= t < n_class   (_fast_hist, metric.py:18-24)
__global__ __launch_bounds__(256) void confusion_hist_kernel(const int32_t* lt, const int32_t* lp, int64_t n, int n_class, unsigned long long* hist) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int t = lt[i], p = lp[i];
        if (t >= 0 && t < n_class && p >= 0 && p < n_class) atomicAdd(&hist[(size_t)n_class * t + p], 1ull);
    }
}
// label = argmax_c uint8(255 * x[c]) over the first C_real channels, first maximum wins (np.argmax), with numpy's
// float -> uint8 cast on x86 (truncate to int32, keep the low 8 bits)   (scores_seg_fake, metric.py:71-77)
template <typename T>
__global__ __launch_bounds__(256) void argmax_u8_kernel(const T* x, int32_t* out, int64_t P, int Cr, int Cp) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        int best = 0, bv = -1;
        for (int c = 0; c < Cr; ++c) {
            int v = (int)(255.f * (float)x[i * Cp + c]) & 0xff;
            if (v > bv) { bv = v; best = c; }
        }
        out[i] = best;
    }
}

static inline int grid_for(int64_t n, int cap = 4096) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

SggTimedLaunch& sgg_timed_launch() {
    static thread_local SggTimedLaunch t;
    return t;
}

extern "C" {

int sgg_version(void) { return SGG_VERSION; }

// ---- measurement hooks (bench.py)
int sgg_event_create(void** ev) {
    if (!ev) return SGG_EINVAL;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return SGG_ELAUNCH; }
    *ev = (void*)e;
    return SGG_OK;
}
int sgg_event_destroy(void* ev) { return ev && hipEventDestroy((hipEvent_t)ev) == hipSuccess ? SGG_OK : SGG_EINVAL; }
int sgg_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (!start || !stop || !ms) return SGG_EINVAL;
    if (hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) { (void)hipGetLastError(); return SGG_ELAUNCH; }
    return SGG_OK;
}
int sgg_time_next_launch(void* start, void* stop) {
    SggTimedLaunch& t = sgg_timed_launch();
    if ((start == nullptr) != (stop == nullptr)) return SGG_EINVAL;
    const int consumed = t.consumed;
    t.start = (hipEvent_t)start; t.stop = (hipEvent_t)stop; t.consumed = 0;
    return consumed;                                     // disarming (NULL, NULL) tells whether the armed pair was used: 1 / 0
}

int sgg_stream_capture_nodes(void* stream, int* n) {
    if (!n) return SGG_EINVAL;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    if (hipStreamGetCaptureInfo_v2((hipStream_t)stream, &st, &id, &graph, &deps, &ndeps) != hipSuccess) { (void)hipGetLastError(); return SGG_ELAUNCH; }
    if (st != hipStreamCaptureStatusActive) { *n = -1; return SGG_OK; }
    size_t total = ndeps;                                // fallback: the capture's current dependency set (0 before the first node)
    if (graph && hipGraphGetNodes(graph, nullptr, &total) != hipSuccess) { (void)hipGetLastError(); total = ndeps; }
    *n = (int)total;
    return SGG_OK;
}

const char* sgg_strerror(int status) {
    switch (status) {
        case SGG_OK: return "SGG_OK";
        case SGG_EINVAL: return "SGG_EINVAL: invalid argument";
        case SGG_EUNSUPPORTED: return "SGG_EUNSUPPORTED: unsupported configuration";
        case SGG_ELAUNCH: return "SGG_ELAUNCH: kernel launch failed";
        case SGG_EWORKSPACE: return "SGG_EWORKSPACE: workspace missing or too small";
        default: return "SGG_E?: unknown status";
    }
}

int sgg_act_fwd(const void* x, void* y, int64_t n, int act, float leak, int dtype, void* stream) {
    return launch_eltwise<0>(x, nullptr, y, n, act, leak, dtype, stream);
}
int sgg_act_bwd(const void* dy, const void* y, void* dx, int64_t n, int act, float leak, int dtype, void* stream) {
    return launch_eltwise<1>(dy, y, dx, n, act, leak, dtype, stream);
}
int sgg_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
    return launch_eltwise<2>(a, b, out, n, 0, 0.f, dtype, stream);
}

size_t sgg_bias_grad_workspace(int64_t P, int C) {
    if (P <= 0 || C <= 0) return 0;
    return (size_t)((P + BG_ROWS - 1) / BG_ROWS) * C * sizeof(float);
}
int sgg_bias_grad(const void* dy, float* db, int64_t P, int C, int C_real, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !db || P <= 0 || C <= 0 || C % SGG_CPAD || C_real <= 0 || C_real > C) return SGG_EINVAL;
    if (!ws || ws_bytes < sgg_bias_grad_workspace(P, C)) return SGG_EWORKSPACE;
    int chunks = (int)((P + BG_ROWS - 1) / BG_ROWS);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SGG_BF16) hipLaunchKernelGGL(colsum_partial_kernel<bf16>, dim3(chunks), dim3(256), 0, s, (const char*)dy, (float*)ws, P, C);
    else if (dtype == SGG_F32) hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(chunks), dim3(256), 0, s, (const char*)dy, (float*)ws, P, C);
    else return SGG_EINVAL;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, (const float*)ws, db, C, C_real, chunks, accumulate, (float*)nullptr);
    return sgg_check_launch();
}
// Grouped call: dy holds the two networks' tensors back to back (P pixels each); db / db2 are their bias gradients.  Same
// arithmetic per network as sgg_bias_grad (bit-identical), two launches instead of four.  ws >= 2 x sgg_bias_grad_workspace(P, C).
int sgg_bias_grad_group2(const void* dy, float* db, float* db2, int64_t P, int C, int C_real, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !db || !db2 || P <= 0 || C <= 0 || C % SGG_CPAD || C_real <= 0 || C_real > C) return SGG_EINVAL;
    if (!ws || ws_bytes < 2 * sgg_bias_grad_workspace(P, C)) return SGG_EWORKSPACE;
    int chunks = (int)((P + BG_ROWS - 1) / BG_ROWS);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SGG_BF16) hipLaunchKernelGGL(colsum_partial_kernel<bf16>, dim3(chunks, 2), dim3(256), 0, s, (const char*)dy, (float*)ws, P, C);
    else if (dtype == SGG_F32) hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(chunks, 2), dim3(256), 0, s, (const char*)dy, (float*)ws, P, C);
    else return SGG_EINVAL;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16, 2), dim3(1024), 0, s, (const float*)ws, db, C, C_real, chunks, accumulate, db2);
    return sgg_check_launch();
}

static bool mask_dims_ok(int N, int hh, int hw, int mh, int mw, int Cr, int Cp) {
    if (N <= 0 || hh <= 0 || hw <= 0 || mh <= 0 || mw <= 0 || Cr <= 0 || Cp < Cr || Cp % SGG_CPAD) return false;
    if (!((hh == mh || hh == 1) && (hw == mw || hw == 1))) return false;   // Keras multiply broadcast rule
    return true;
}
int sgg_mask_reduce_fwd(const void* h4, const float* mask, float* out, int N, int hh, int hw, int mh, int mw, int Cr, int Cp, int dtype, void* stream) {
    if (!h4 || !mask || !out || !mask_dims_ok(N, hh, hw, mh, mw, Cr, Cp)) return SGG_EINVAL;
    int total = N * mh * mw;
    if (dtype == SGG_BF16) hipLaunchKernelGGL(mask_reduce_fwd_kernel<bf16>, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, (const bf16*)h4, mask, out, N, hh, hw, mh, mw, Cr, Cp);
    else if (dtype == SGG_F32) hipLaunchKernelGGL(mask_reduce_fwd_kernel<float>, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, (const float*)h4, mask, out, N, hh, hw, mh, mw, Cr, Cp);
    else return SGG_EINVAL;
    return sgg_check_launch();
}
int sgg_mask_reduce_bwd(const float* dout, const float* mask, void* dh4, int N, int hh, int hw, int mh, int mw, int Cr, int Cp, int dtype, void* stream) {
    if (!dout || !mask || !dh4 || !mask_dims_ok(N, hh, hw, mh, mw, Cr, Cp)) return SGG_EINVAL;
    int total = N * hh * hw * Cp;
    if (dtype == SGG_BF16) hipLaunchKernelGGL(mask_reduce_bwd_kernel<bf16>, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, dout, mask, (bf16*)dh4, N, hh, hw, mh, mw, Cr, Cp);
    else if (dtype == SGG_F32) hipLaunchKernelGGL(mask_reduce_bwd_kernel<float>, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, dout, mask, (float*)dh4, N, hh, hw, mh, mw, Cr, Cp);
    else return SGG_EINVAL;
    return sgg_check_launch();
}

int sgg_bce_logits(const float* logits, int64_t n, float label, float weight, float gscale, float* loss, float* dlogits, int accumulate, void* stream) {
    if (!logits || n <= 0) return SGG_EINVAL;
    hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, n, label, weight, gscale, loss, dlogits, accumulate);
    return sgg_check_launch();
}

int sgg_mse_const(const float* x, int64_t n, float target, float weight, float gscale, float* loss, float* dx, int accumulate, void* stream) {
    if (!x || n <= 0) return SGG_EINVAL;
    hipLaunchKernelGGL(mse_const_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, target, weight, gscale, loss, dx, accumulate);
    return sgg_check_launch();
}

int sgg_seg_edge_weight(const void* seg, float* out, int N, int H, int W, int C_real, int Cpad, int dtype, void* stream) {
    if (!seg || !out || N <= 0 || H < 2 || W < 2 || C_real <= 0 || Cpad < C_real) return SGG_EINVAL;
    int64_t total = (int64_t)N * H * W;
    if (dtype == SGG_BF16) hipLaunchKernelGGL(seg_edge_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16*)seg, out, N, H, W, C_real, Cpad);
    else if (dtype == SGG_F32) hipLaunchKernelGGL(seg_edge_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)seg, out, N, H, W, C_real, Cpad);
    else return SGG_EINVAL;
    return sgg_check_launch();
}

#define GL_BLOCKS 1024
size_t sgg_gradloss_workspace(int N, int H, int W, int C_real) {
    if (N <= 0 || H <= 0 || W <= 0 || C_real <= 0) return 0;
    return (size_t)N * H * W * C_real * 2 * sizeof(float) + GL_BLOCKS * sizeof(float);
}
int sgg_gradloss(const void* in, const void* target, const float* weight, int N, int H, int W, int C_real, int Cpad, float lambda,
                 float gscale, float* loss, void* din, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !target || !weight || !loss || N <= 0 || H <= 0 || W <= 0 || C_real <= 0 || Cpad < C_real) return SGG_EINVAL;
    if (!ws || ws_bytes < sgg_gradloss_workspace(N, H, W, C_real)) return SGG_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)ws;
    float* coef = din ? partial + GL_BLOCKS : nullptr;
    int64_t total = (int64_t)N * H * W;
    int blocks = (int)((total + 255) / 256); if (blocks > GL_BLOCKS) blocks = GL_BLOCKS;
    double denom = (double)total * (2.0 * C_real);               // mean over 2C derivative channels, then over pixels
    float gval = (float)((double)lambda * gscale / denom);
    if (dtype != SGG_BF16 && dtype != SGG_F32) return SGG_EINVAL;
    const bool vec = C_real <= 4 && Cpad % SGG_CPAD == 0;        // the real channels sit in the pixel's first 16-byte vector
    if (vec && dtype == SGG_BF16) hipLaunchKernelGGL(gradloss_fwd_vec_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const char*)in, (const char*)target, weight, coef, partial, N, H, W, C_real, Cpad, gval);
    else if (vec) hipLaunchKernelGGL(gradloss_fwd_vec_kernel<float>, dim3(blocks), dim3(256), 0, s, (const char*)in, (const char*)target, weight, coef, partial, N, H, W, C_real, Cpad, gval);
    else if (dtype == SGG_BF16) hipLaunchKernelGGL(gradloss_fwd_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)in, (const bf16*)target, weight, coef, partial, N, H, W, C_real, Cpad, gval);
    else hipLaunchKernelGGL(gradloss_fwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)in, (const float*)target, weight, coef, partial, N, H, W, C_real, Cpad, gval);
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, s, (const float*)partial, blocks, (double)lambda / denom, loss, accumulate & 1);
    if (din) {
        int64_t tot2 = total * Cpad;
        if (vec && dtype == SGG_BF16) hipLaunchKernelGGL(gradloss_bwd_vec_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)coef, (char*)din, N, H, W, C_real, Cpad, (accumulate >> 1) & 1);
        else if (vec) hipLaunchKernelGGL(gradloss_bwd_vec_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)coef, (char*)din, N, H, W, C_real, Cpad, (accumulate >> 1) & 1);
        else if (dtype == SGG_BF16) hipLaunchKernelGGL(gradloss_bwd_kernel<bf16>, dim3(grid_for(tot2)), dim3(256), 0, s, (const float*)coef, (bf16*)din, N, H, W, C_real, Cpad, (accumulate >> 1) & 1);
        else hipLaunchKernelGGL(gradloss_bwd_kernel<float>, dim3(grid_for(tot2)), dim3(256), 0, s, (const float*)coef, (float*)din, N, H, W, C_real, Cpad, (accumulate >> 1) & 1);
    }
    return sgg_check_launch();
}

size_t sgg_l1_loss_workspace(int64_t P, int Cpad) {
    if (P <= 0 || Cpad <= 0) return 0;
    int64_t nvec = P * Cpad / 4;      // worst case (f32)
    return (size_t)((nvec + L1_ROWS - 1) / L1_ROWS) * sizeof(float);
}
int sgg_l1_loss(const void* a, const void* b, int64_t P, int Cr, int Cp, float weight, float gscale, float* loss, void* db,
                int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!a || !b || !loss || P <= 0 || Cr <= 0 || Cp < Cr || Cp % SGG_CPAD) return SGG_EINVAL;
    if (!ws || ws_bytes < sgg_l1_loss_workspace(P, Cp)) return SGG_EWORKSPACE;
    int vec = dtype == SGG_BF16 ? 8 : 4;
    int64_t nvec = P * Cp / vec;
    int chunks = (int)((nvec + L1_ROWS - 1) / L1_ROWS);
    double cnt = (double)P * Cr;
    float gval = (float)((double)weight * gscale / cnt);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SGG_BF16) hipLaunchKernelGGL(l1_partial_kernel<bf16>, dim3(chunks), dim3(256), 0, s, (const char*)a, (const char*)b, (char*)db, (float*)ws, nvec, Cr, Cp, gval, (accumulate >> 1) & 1);
    else if (dtype == SGG_F32) hipLaunchKernelGGL(l1_partial_kernel<float>, dim3(chunks), dim3(256), 0, s, (const char*)a, (const char*)b, (char*)db, (float*)ws, nvec, Cr, Cp, gval, (accumulate >> 1) & 1);
    else return SGG_EINVAL;
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, chunks, (double)weight / cnt, loss, accumulate & 1);
    return sgg_check_launch();
}

int sgg_adam(float* theta, const float* g, float* m, float* v, int64_t n, int t, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
    if (!theta || !g || !m || !v || n < 0 || t < 1) return SGG_EINVAL;
    if (n == 0) return SGG_OK;
    double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, theta, g, m, v, n, (float)lr_t, beta1, beta2, eps, grad_scale);
    return sgg_check_launch();
}

int sgg_adam_iter(float* theta, const float* g, float* m, float* v, int64_t n, int64_t* state, float lr, float beta1, float beta2,
                  float eps, float grad_scale, void* stream) {
    if (!theta || !g || !m || !v || !state || n < 0) return SGG_EINVAL;
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, lr, beta1, beta2);
    int rc = sgg_check_launch();
    if (rc || n == 0) return rc;
    hipLaunchKernelGGL(adam_iter_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, theta, g, m, v, n, (const int64_t*)state,
                       beta1, beta2, eps, grad_scale);
    return sgg_check_launch();
}

int sgg_seg_class_table(uint32_t* keys_host, uint8_t* vals_host, int capacity) {
    if (!keys_host || !vals_host || capacity < 21) return SGG_EINVAL;
    for (int i = 0; i < 21; ++i) { keys_host[i] = hSegKeys[i]; vals_host[i] = hSegVals[i]; }
    return 21;
}

int sgg_seg_class_map(const uint8_t* rgb, int channels, int64_t n_pixels, uint8_t* out, void* stream) {
    if (n_pixels == 0) return SGG_OK;          // empty image: nothing to do (pointers may be null)
    if (!rgb || !out || channels < 3 || n_pixels < 0) return SGG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int64_t done = 0;
    if ((channels == 3 || channels == 4) && n_pixels >= 4 && (((uintptr_t)rgb | (uintptr_t)out) & 3) == 0) {
        const int64_t groups = n_pixels / 4;
        if (channels == 3) hipLaunchKernelGGL(seg_class_vec4_kernel<3>, dim3(grid_for(groups, 2048)), dim3(256), 0, s, (const uint32_t*)rgb, groups, (uint32_t*)out);
        else hipLaunchKernelGGL(seg_class_vec4_kernel<4>, dim3(grid_for(groups, 2048)), dim3(256), 0, s, (const uint32_t*)rgb, groups, (uint32_t*)out);
        done = groups * 4;
    }
    if (done < n_pixels)
        hipLaunchKernelGGL(seg_class_kernel, dim3(grid_for(n_pixels - done, 2048)), dim3(256), 0, s, rgb + done * channels, channels, n_pixels - done, out + done);
    return sgg_check_launch();
}

int sgg_onehot_resample(const uint8_t* idx, float* mask, int N, int H, int W, int oh, int ow, int n_classes, void* stream) {
    if (!idx || !mask || N <= 0 || H <= 0 || W <= 0 || oh <= 0 || ow <= 0 || n_classes <= 0) return SGG_EINVAL;
    int64_t total = (int64_t)N * oh * ow * n_classes;
    hipLaunchKernelGGL(onehot_resample_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, idx, mask, N, H, W, oh, ow, n_classes);
    return sgg_check_launch();
}

int sgg_confusion_hist(const int32_t* label_true, const int32_t* label_pred, int64_t n, int n_class, uint64_t* hist, void* stream) {
    if (!label_true || !label_pred || !hist || n < 0 || n_class <= 0) return SGG_EINVAL;
    if (n == 0) return SGG_OK;
    hipLaunchKernelGGL(confusion_hist_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, label_true, label_pred, n, n_class, (unsigned long long*)hist);
    return sgg_check_launch();
}
int sgg_argmax_u8_labels(const void* x, int32_t* labels, int64_t P, int C_real, int Cpad, int dtype, void* stream) {
    if (!x || !labels || P <= 0 || C_real <= 0 || Cpad < C_real) return SGG_EINVAL;
    if (dtype == SGG_BF16) hipLaunchKernelGGL(argmax_u8_kernel<bf16>, dim3(grid_for(P)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, labels, P, C_real, Cpad);
    else if (dtype == SGG_F32) hipLaunchKernelGGL(argmax_u8_kernel<float>, dim3(grid_for(P)), dim3(256), 0, (hipStream_t)stream, (const float*)x, labels, P, C_real, Cpad);
    else return SGG_EINVAL;
    return sgg_check_launch();
}

int sgg_pad_channels(const float* src, void* dst, int64_t P, int Cs, int Cd, int dtype, void* stream) {
    if (!src || !dst || P <= 0 || Cs <= 0 || Cd < Cs) return SGG_EINVAL;
    if (dtype == SGG_BF16) hipLaunchKernelGGL(pad_channels_kernel<bf16>, dim3(grid_for(P * Cd)), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, P, Cs, Cd);
    else if (dtype == SGG_F32) hipLaunchKernelGGL(pad_channels_kernel<float>, dim3(grid_for(P * Cd)), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, P, Cs, Cd);
    else return SGG_EINVAL;
    return sgg_check_launch();
}
int sgg_unpad_channels(const void* src, float* dst, int64_t P, int Cs, int Cd, int dtype, void* stream) {
    if (!src || !dst || P <= 0 || Cd <= 0 || Cs < Cd) return SGG_EINVAL;
    if (dtype == SGG_BF16) hipLaunchKernelGGL(unpad_channels_kernel<bf16>, dim3(grid_for(P * Cd)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, dst, P, Cs, Cd);
    else if (dtype == SGG_F32) hipLaunchKernelGGL(unpad_channels_kernel<float>, dim3(grid_for(P * Cd)), dim3(256), 0, (hipStream_t)stream, (const float*)src, dst, P, Cs, Cd);
    else return SGG_EINVAL;
    return sgg_check_launch();
}

}  // extern "C"
