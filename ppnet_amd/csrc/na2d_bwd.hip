// na2d_bwd.hip — backward of the 2-D neighbourhood attention (the training step, reference GenNet/train.py:93-147 /
// SegNet/mmseg/apis/train.py:67-167; NATTEN's natten2dqkrpb / natten2dav backward kernels behind SegNet/nat.py:14,111-120).  Same
// semantics as the forward kernel (oracle/na_np.py): kernel 7, head dim 32, dilation d, window start clamp(u - 3, 0, n - 7) inside the
// query's dilation group, relative position bias rpb[h][ki + ws - u + 6][kj + ws' - v + 6].  float32 arithmetic, float32 or bfloat16
// tensors.
//
// Two passes over 8 x 8 regions of a dilation group's sub-image, both tiled through LDS, the probabilities RECOMPUTED in the second
// (what crosses between them is the softmax statistics of a query, 16 bytes, instead of NATTEN's `attn` and its gradient, 392):
//   query pass   the region's queries against the staged K / V halo (<= 14 x 14 keys): S, P, dP = dout . v, D = sum P dP,
//                dS = P (dP - D);  dq = scale * sum dS k;  (max, 1 / sum, D) -> stats;  the region's share of drpb, summed in a fixed
//                order (bin by bin over the region's queries) -> partial sums per workgroup
//   key pass     the region's keys against the staged q / dout / stats of the queries whose window holds them (the inverse of the
//                clamped-window map: a contiguous range per axis, <= 17 x 17 queries): p = exp(s - max) / sum, dS as above;
//                dk = scale * sum dS q,  dv = sum p dout.  No atomics anywhere.
//   drpb         the workgroups' partial sums added in workgroup order: the gradient is bit-reproducible.
// Four lanes share a query (key): lane r holds channels 8r .. 8r+7 of its q / dout / dq (k / v / dk / dv), a dot product is 8 multiply-
// adds and two quad exchanges, and nothing is reduced at the end.
#include <hip/hip_bf16.h>
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

namespace {
constexpr int BK = 7, BN = 3, BHD = 32;
constexpr int TR = 8;                    // region side
constexpr int QH = TR + 6;               // key halo of a region of queries
constexpr int KH = TR + 9;               // query halo of a region of keys (a border key is seen by up to 10 queries per axis)
constexpr int NTHR = TR * TR * 4;

__device__ __forceinline__ int clampw(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// 8 channels of a token's row -> float
__device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void load8(const __hip_bfloat16* p, float (&o)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(__hip_bfloat16* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
__device__ __forceinline__ void lds8(const float* p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ float dot8(const float (&a)[8], const float (&b)[8]) {
    float s = a[0] * b[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) s = fmaf(a[c], b[c], s);
    return s;
}
// sum over the 4 lanes of a quad (every lane gets it; the order is the same on all of them)
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    return v;
}
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 64));
    v = fmaxf(v, __shfl_xor(v, 2, 64));
    return v;
}

// the region of a workgroup: image b, dilation group (gi, gj) with its hs x ws sub-image, region origin (ty0, tx0)
struct Region { int b, gi, gj, hs, ws, ty0, tx0; bool any; };
__device__ __forceinline__ Region region_of(int wg, int H, int W, int dil, int tiles_y, int tiles_x) {
    Region r;
    const int ntiles = tiles_y * tiles_x;
    const int bz = wg / ntiles, tile = wg - bz * ntiles;
    r.b = bz / (dil * dil);
    const int g2 = bz - r.b * dil * dil;
    r.gi = g2 / dil; r.gj = g2 - r.gi * dil;
    r.hs = (H - r.gi + dil - 1) / dil; r.ws = (W - r.gj + dil - 1) / dil;
    r.ty0 = (tile / tiles_x) * TR; r.tx0 = (tile % tiles_x) * TR;
    r.any = r.ty0 < r.hs && r.tx0 < r.ws;                                  // groups differ by one row / column
    return r;
}
}  // namespace

// LDS: K halo [QH*QH][32] f32 | V halo | dS [64][49] | rpb[h] 169
template <typename T>
__global__ __launch_bounds__(NTHR) void na2d_bwd_query_kernel(const T* __restrict__ qkv, const float* __restrict__ rpb, const T* __restrict__ dout,
                                                              T* __restrict__ dqkv, float4* __restrict__ stats, float* __restrict__ partial, int H,
                                                              int W, int heads, int dil, float scale, int tiles_y, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Kh = sm;
    float* Vh = Kh + QH * QH * BHD;
    float* dS = Vh + QH * QH * BHD;
    float* rp = dS + TR * TR * BK * BK;
    const int h = blockIdx.y, tid = threadIdx.x;
    float* my_partial = partial + ((size_t)h * gridDim.x + blockIdx.x) * 169;
    const Region R = region_of(blockIdx.x, H, W, dil, tiles_y, tiles_x);
    if (!R.any) {                                                          // workgroup-uniform
        if (tid < 169) my_partial[tid] = 0.f;
        return;
    }
    const size_t tok = (size_t)3 * heads * BHD;
    const int R0 = clampw(R.ty0 - BN, 0, R.hs - BK), C0 = clampw(R.tx0 - BN, 0, R.ws - BK);
    const int NR = clampw(min(R.ty0 + TR - 1, R.hs - 1) - BN, 0, R.hs - BK) + BK - R0;
    const int NC = clampw(min(R.tx0 + TR - 1, R.ws - 1) - BN, 0, R.ws - BK) + BK - C0;
    // ---- stage the K and V halo: a piece = 8 channels of a key
    for (int p = tid; p < NR * NC * 4; p += NTHR) {
        const int slot = p >> 2, c8 = (p & 3) * 8;
        const int t = slot / NC, sc = slot - t * NC;
        const int y = R.gi + (R0 + t) * dil, x = R.gj + (C0 + sc) * dil;
        const T* row = qkv + ((size_t)(R.b * H + y) * W + x) * tok + (size_t)h * BHD + c8;
        float kk[8], vv[8];
        load8(row + (size_t)heads * BHD, kk);
        load8(row + (size_t)2 * heads * BHD, vv);
        store8(Kh + (t * QH + sc) * BHD + c8, kk);
        store8(Vh + (t * QH + sc) * BHD + c8, vv);
    }
    if (tid < 169) rp[tid] = rpb[(size_t)h * 169 + tid];
    // ---- this lane: channels 8r .. 8r+7 of query ql of the region
    const int ql = tid >> 2, r = tid & 3;
    const int u = R.ty0 + (ql >> 3), v = R.tx0 + (ql & 7);
    const bool qvalid = u < R.hs && v < R.ws;
    const int uc = min(u, R.hs - 1), vc = min(v, R.ws - 1);                // dead queries shadow a live one (never stored)
    const int wi = clampw(uc - BN, 0, R.hs - BK), wj = clampw(vc - BN, 0, R.ws - BK);
    const int y = R.gi + uc * dil, x = R.gj + vc * dil;
    float q[8], g[8];
    load8(qkv + ((size_t)(R.b * H + y) * W + x) * tok + (size_t)h * BHD + 8 * r, q);
    load8(dout + ((size_t)(R.b * H + y) * W + x) * ((size_t)heads * BHD) + (size_t)h * BHD + 8 * r, g);
    __syncthreads();

    // ---- pass 1: logits and dP of the 49 keys; key n stays with lane n % 4 of the quad
    const float* kbase = Kh + ((wi - R0) * QH + (wj - C0)) * BHD + 8 * r;
    const float* vbase = Vh + ((wi - R0) * QH + (wj - C0)) * BHD + 8 * r;
    const float* rbase = rp + (wi - uc + BK - 1) * 13 + (wj - vc + BK - 1);
    float s_own[13], dp_own[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) { s_own[i] = -3.0e38f; dp_own[i] = 0.f; }
#pragma unroll
    for (int n = 0; n < BK * BK; ++n) {
        const int ki = n / BK, kj = n - ki * BK;
        float kk[8], vv[8];
        lds8(kbase + (ki * QH + kj) * BHD, kk);
        lds8(vbase + (ki * QH + kj) * BHD, vv);
        const float s = quad_sum(dot8(q, kk)) * scale + rbase[ki * 13 + kj];
        const float d = quad_sum(dot8(g, vv));
        if ((n & 3) == r) { s_own[n >> 2] = s; dp_own[n >> 2] = d; }
    }
    float mx = s_own[0];
#pragma unroll
    for (int i = 1; i < 13; ++i) mx = fmaxf(mx, s_own[i]);
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 13; ++i) { s_own[i] = __expf(s_own[i] - mx); sum += s_own[i]; }    // (the 3 empty slots: exp(-huge) = 0)
    sum = quad_sum(sum);
    const float inv = 1.0f / sum;
    float dsum = 0.f;
#pragma unroll
    for (int i = 0; i < 13; ++i) { s_own[i] *= inv; dsum = fmaf(s_own[i], dp_own[i], dsum); }
    dsum = quad_sum(dsum);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int n = 4 * i + r;
        if (n < BK * BK) dS[ql * (BK * BK) + n] = qvalid ? s_own[i] * (dp_own[i] - dsum) : 0.f;
    }
    if (r == 0 && qvalid) stats[((size_t)(R.b * heads + h) * H + y) * W + x] = make_float4(mx, inv, dsum, 0.f);
    __syncthreads();

    // ---- pass 2: dq = scale * sum dS k
    float dq[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) dq[c] = 0.f;
#pragma unroll
    for (int n = 0; n < BK * BK; ++n) {
        const int ki = n / BK, kj = n - ki * BK;
        float kk[8];
        lds8(kbase + (ki * QH + kj) * BHD, kk);
        const float ds = dS[ql * (BK * BK) + n];
#pragma unroll
        for (int c = 0; c < 8; ++c) dq[c] = fmaf(ds, kk[c], dq[c]);
    }
    if (qvalid) {
#pragma unroll
        for (int c = 0; c < 8; ++c) dq[c] *= scale;
        store8(dqkv + ((size_t)(R.b * H + y) * W + x) * tok + (size_t)h * BHD + 8 * r, dq);
    }
    // ---- the region's share of drpb: bin (a, b) collects dS of key (a - 6 + u - wi, b - 6 + v - wj) of every query that has one
    if (tid < 169) {
        const int a = tid / 13, b = tid - a * 13;
        float acc = 0.f;
        for (int qq = 0; qq < TR * TR; ++qq) {
            const int uu = min(R.ty0 + (qq >> 3), R.hs - 1), vv = min(R.tx0 + (qq & 7), R.ws - 1);
            const int ki = a - (BK - 1) + uu - clampw(uu - BN, 0, R.hs - BK), kj = b - (BK - 1) + vv - clampw(vv - BN, 0, R.ws - BK);
            if (ki >= 0 && ki < BK && kj >= 0 && kj < BK) acc += dS[qq * (BK * BK) + ki * BK + kj];
        }
        my_partial[tid] = acc;
    }
}

// LDS: q halo [KH*KH][32] f32 | dout halo | stats [KH*KH] float4 | rpb[h] 169
template <typename T>
__global__ __launch_bounds__(NTHR) void na2d_bwd_key_kernel(const T* __restrict__ qkv, const float* __restrict__ rpb, const T* __restrict__ dout,
                                                            T* __restrict__ dqkv, const float4* __restrict__ stats, int H, int W, int heads, int dil,
                                                            float scale, int tiles_y, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Qh = sm;
    float* Gh = Qh + KH * KH * BHD;
    float4* St = reinterpret_cast<float4*>(Gh + KH * KH * BHD);
    float* rp = reinterpret_cast<float*>(St + KH * KH);
    const int h = blockIdx.y, tid = threadIdx.x;
    const Region R = region_of(blockIdx.x, H, W, dil, tiles_y, tiles_x);
    if (!R.any) return;                                                    // workgroup-uniform
    const size_t tok = (size_t)3 * heads * BHD;
    // the queries whose window holds key i: [i <= 6 ? 0 : i - 3, i >= hs - 7 ? hs - 1 : i + 3] (both ends monotone in i)
    const int ilast = min(R.ty0 + TR - 1, R.hs - 1), jlast = min(R.tx0 + TR - 1, R.ws - 1);
    const int qr0 = R.ty0 <= BK - 1 ? 0 : R.ty0 - BN, qr1 = ilast >= R.hs - BK ? R.hs - 1 : ilast + BN;
    const int qc0 = R.tx0 <= BK - 1 ? 0 : R.tx0 - BN, qc1 = jlast >= R.ws - BK ? R.ws - 1 : jlast + BN;
    const int NQR = qr1 - qr0 + 1, NQC = qc1 - qc0 + 1;                    // <= KH each
    for (int p = tid; p < NQR * NQC * 4; p += NTHR) {
        const int slot = p >> 2, c8 = (p & 3) * 8;
        const int t = slot / NQC, sc = slot - t * NQC;
        const int y = R.gi + (qr0 + t) * dil, x = R.gj + (qc0 + sc) * dil;
        float qq[8], gg[8];
        load8(qkv + ((size_t)(R.b * H + y) * W + x) * tok + (size_t)h * BHD + c8, qq);
        load8(dout + ((size_t)(R.b * H + y) * W + x) * ((size_t)heads * BHD) + (size_t)h * BHD + c8, gg);
        store8(Qh + (t * KH + sc) * BHD + c8, qq);
        store8(Gh + (t * KH + sc) * BHD + c8, gg);
        if (c8 == 0) St[t * KH + sc] = stats[((size_t)(R.b * heads + h) * H + y) * W + x];
    }
    if (tid < 169) rp[tid] = rpb[(size_t)h * 169 + tid];
    // ---- this lane: channels 8r .. 8r+7 of key kl of the region
    const int kl = tid >> 2, r = tid & 3;
    const int i0 = R.ty0 + (kl >> 3), j0 = R.tx0 + (kl & 7);
    const bool kvalid = i0 < R.hs && j0 < R.ws;
    const int i = min(i0, R.hs - 1), j = min(j0, R.ws - 1);
    const int y = R.gi + i * dil, x = R.gj + j * dil;
    const T* row = qkv + ((size_t)(R.b * H + y) * W + x) * tok + (size_t)h * BHD + 8 * r;
    float k[8], vv[8], dk[8], dv[8];
    load8(row + (size_t)heads * BHD, k);
    load8(row + (size_t)2 * heads * BHD, vv);
#pragma unroll
    for (int c = 0; c < 8; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
    __syncthreads();
    const int ulo = i <= BK - 1 ? 0 : i - BN, uhi = i >= R.hs - BK ? R.hs - 1 : i + BN;
    const int vlo = j <= BK - 1 ? 0 : j - BN, vhi = j >= R.ws - BK ? R.ws - 1 : j + BN;
    for (int u = ulo; u <= uhi; ++u) {
        const int wi = clampw(u - BN, 0, R.hs - BK);
        if (i < wi || i > wi + BK - 1) continue;
        for (int v = vlo; v <= vhi; ++v) {
            const int wj = clampw(v - BN, 0, R.ws - BK);
            if (j < wj || j > wj + BK - 1) continue;
            const int slot = (u - qr0) * KH + (v - qc0);
            float qq[8], gg[8];
            lds8(Qh + slot * BHD + 8 * r, qq);
            lds8(Gh + slot * BHD + 8 * r, gg);
            const float4 st = St[slot];                                    // max, 1 / sum, D of query (u, v)
            const float s = quad_sum(dot8(k, qq)) * scale + rp[(i - u + BK - 1) * 13 + (j - v + BK - 1)];
            const float dp = quad_sum(dot8(vv, gg));
            const float p = __expf(s - st.x) * st.y;
            const float ds = p * (dp - st.z);
#pragma unroll
            for (int c = 0; c < 8; ++c) { dk[c] = fmaf(ds, qq[c], dk[c]); dv[c] = fmaf(p, gg[c], dv[c]); }
        }
    }
    if (kvalid) {
#pragma unroll
        for (int c = 0; c < 8; ++c) dk[c] *= scale;
        T* orow = dqkv + ((size_t)(R.b * H + y) * W + x) * tok + (size_t)h * BHD + 8 * r;
        store8(orow + (size_t)heads * BHD, dk);
        store8(orow + (size_t)2 * heads * BHD, dv);
    }
}

// drpb[h][bin] = the workgroups' partial sums: slice s of a bin adds workgroups s, s + 6, ... in order, the 6 slices are added in
// order (a fixed association: bit-reproducible)
__global__ __launch_bounds__(1024) void na2d_bwd_rpb_kernel(const float* __restrict__ partial, float* __restrict__ drpb, int nwg) {
    __shared__ float part[6][169];
    const int h = blockIdx.x, t = threadIdx.x % 169, s = threadIdx.x / 169;
    if (s < 6) {
        const float* p = partial + (size_t)h * nwg * 169 + t;
        float a = 0.f;
        for (int w = s; w < nwg; w += 6) a += p[(size_t)w * 169];
        part[s][t] = a;
    }
    __syncthreads();
    if (threadIdx.x < 169) drpb[(size_t)h * 169 + t] = ((part[0][t] + part[1][t]) + (part[2][t] + part[3][t])) + (part[4][t] + part[5][t]);
}

static void bwd_geometry(int H, int W, int dil, int B, int& tiles_y, int& tiles_x, long long& nwg) {
    const int hs = (H + dil - 1) / dil, ws = (W + dil - 1) / dil;            // largest sub-image
    tiles_y = (hs + TR - 1) / TR; tiles_x = (ws + TR - 1) / TR;
    nwg = (long long)tiles_y * tiles_x * B * dil * dil;
}

// floats of workspace the backward needs: a float4 of softmax statistics per (query, head), 169 partial sums per (workgroup, head)
long long na2d_bwd_workspace_floats(int B, int H, int W, int heads, int dil) {
    int ty, tx; long long nwg;
    bwd_geometry(H, W, dil, B, ty, tx, nwg);
    return (long long)B * heads * H * W * 4 + nwg * heads * 169;
}

template <typename T>
static int bwd_typed(const void* qkv, const float* rpb, const void* dout, void* dqkv, float* drpb, float* ws, int B, int H, int W, int heads, int dil,
                     float scale, hipStream_t stream) {
    int tiles_y, tiles_x; long long nwg;
    bwd_geometry(H, W, dil, B, tiles_y, tiles_x, nwg);
    if (nwg >= (1LL << 31) || heads > 65535) return -1;
    float4* stats = reinterpret_cast<float4*>(ws);
    float* partial = ws + (size_t)B * heads * H * W * 4;
    constexpr int LDS_Q = (2 * QH * QH * BHD + TR * TR * BK * BK + 176) * 4, LDS_K = (2 * KH * KH * BHD + KH * KH * 4 + 176) * 4;
    static DeviceOnce attr_q, attr_k;
    if (const int e = dynamic_lds_once(attr_q, (const void*)na2d_bwd_query_kernel<T>, LDS_Q)) return e;
    if (const int e = dynamic_lds_once(attr_k, (const void*)na2d_bwd_key_kernel<T>, LDS_K)) return e;
    const dim3 grid((unsigned)nwg, heads);
    hipLaunchKernelGGL((na2d_bwd_query_kernel<T>), grid, dim3(NTHR), LDS_Q, stream, (const T*)qkv, rpb, (const T*)dout, (T*)dqkv, stats, partial, H, W,
                       heads, dil, scale, tiles_y, tiles_x);
    hipLaunchKernelGGL((na2d_bwd_key_kernel<T>), grid, dim3(NTHR), LDS_K, stream, (const T*)qkv, rpb, (const T*)dout, (T*)dqkv, (const float4*)stats, H, W,
                       heads, dil, scale, tiles_y, tiles_x);
    hipLaunchKernelGGL(na2d_bwd_rpb_kernel, dim3(heads), dim3(1024), 0, stream, (const float*)partial, drpb, (int)nwg);
    return (int)hipGetLastError();
}

int na2d_bwd_launch(const void* qkv, const float* rpb, const void* dout, void* dqkv, float* drpb, float* ws, int B, int H, int W, int heads, int dil,
                    float scale, int dtype, hipStream_t stream) {
    return dtype == 0 ? bwd_typed<float>(qkv, rpb, dout, dqkv, drpb, ws, B, H, W, heads, dil, scale, stream)
                      : bwd_typed<__hip_bfloat16>(qkv, rpb, dout, dqkv, drpb, ws, B, H, W, heads, dil, scale, stream);
}

}  // namespace ppn
