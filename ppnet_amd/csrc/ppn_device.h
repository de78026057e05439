// ppn_device.h — device helpers shared by the gfx950 kernels: Philox4x32-10 draws, wave64 /
// workgroup reductions, strict (unfused) double arithmetic.
//
// All kernels in this library are compiled with -ffp-contract=off: the parity contract with
// the oracle (oracle/edage_np.py, ARITH="plain") is rn(a*b) then rn(+), never an fma.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PPN_WAVE 64
#define PPN_TIE_EPS 1e-9

namespace ppn {

// two float32 -> one dword of two bfloat16 (lo in bits 0..15), round-to-nearest-even.  A plain cast, which hipcc lowers to
// v_cvt_pk_bf16_f32: a NaN stays a NaN (integer rounding on the float bits turns some NaNs into 0 or infinity,
// MI355X_MICROARCH.md "Correctness boundaries").
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}


// ------------------------------------------------------------------------------------ Philox
// Salmon et al. SC'11, Random123 constants; counter = (block, inst_hi, inst_lo, stream).
enum : uint32_t { STREAM_PATH = 1, STREAM_POCKET = 2, STREAM_PLACE = 3, STREAM_OBST = 4 };

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r) { k0 += W0; k1 += W1; }
        uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    }
    return {c0, c1, c2, c3};
}

__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
    // numpy random_sample recipe: (a>>5, b>>6) -> (a*2^26 + b) / 2^53; every step exact
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// draw index d of (stream, instance): block d>>1, even -> words (0,1), odd -> (2,3)
__device__ __forceinline__ double philox_double(uint64_t seed, uint32_t stream, uint64_t inst, uint32_t d) {
    u32x4 o = philox4x32_10(d >> 1, (uint32_t)(inst >> 32), (uint32_t)inst, stream,
                            (uint32_t)seed, (uint32_t)(seed >> 32));
    return (d & 1) ? u53(o.z, o.w) : u53(o.x, o.y);
}

// both doubles of block b (draws 2b and 2b+1)
__device__ __forceinline__ void philox_double2(uint64_t seed, uint32_t stream, uint64_t inst, uint32_t b,
                                               double& d0, double& d1) {
    u32x4 o = philox4x32_10(b, (uint32_t)(inst >> 32), (uint32_t)inst, stream,
                            (uint32_t)seed, (uint32_t)(seed >> 32));
    d0 = u53(o.x, o.y);
    d1 = u53(o.z, o.w);
}

// torch.rand recipe: (w & 0xFFFFFF) / 2^24; draw d: block d>>2, word d&3
__device__ __forceinline__ float philox_float(uint64_t seed, uint32_t stream, uint64_t inst, uint32_t d) {
    u32x4 o = philox4x32_10(d >> 2, (uint32_t)(inst >> 32), (uint32_t)inst, stream,
                            (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t w = (d & 3) == 0 ? o.x : (d & 3) == 1 ? o.y : (d & 3) == 2 ? o.z : o.w;
    return (float)(w & 0xFFFFFFu) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;   // valid in lane 0
}

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;   // valid in every lane
}

struct MinIdx { double v; int i; };

// (value, index) lexicographic minimum: smallest value, then smallest index ("first strict minimum")
__device__ __forceinline__ MinIdx min_idx(MinIdx a, MinIdx b) {
    return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

__device__ __forceinline__ MinIdx wave_min_idx(MinIdx m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MinIdx t;
        t.v = __shfl_xor(m.v, o, 64);
        t.i = __shfl_xor(m.i, o, 64);
        m = min_idx(m, t);
    }
    return m;
}

// Workgroup-wide reductions through a small LDS scratch (one slot per wave).  Every thread of
// the workgroup must call; the result is returned to every thread.  NW = waves per workgroup.
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += scratch[i];   // fixed order -> deterministic
    return s;
}

template <int NW>
__device__ __forceinline__ double block_min(double v, double* scratch) {
    v = wave_min(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double s = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) s = fmin(s, scratch[i]);
    return s;
}

template <int NW>
__device__ __forceinline__ double block_max(double v, double* scratch) {
    return -block_min<NW>(-v, scratch);
}

template <int NW>
__device__ __forceinline__ MinIdx block_min_idx(MinIdx m, double* scratch_v, int* scratch_i) {
    m = wave_min_idx(m);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { scratch_v[w] = m.v; scratch_i[w] = m.i; }
    __syncthreads();
    MinIdx r{scratch_v[0], scratch_i[0]};
#pragma unroll
    for (int i = 1; i < NW; ++i) r = min_idx(r, MinIdx{scratch_v[i], scratch_i[i]});
    return r;
}

template <int NW>
__device__ __forceinline__ int block_min_int(int v, int* scratch_i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch_i[w] = v;
    __syncthreads();
    int r = scratch_i[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = min(r, scratch_i[i]);
    return r;
}

// ------------------------------------------------------------------------------------ geometry
// coord_rotation (EDaGe-PP/Path.py:271-274), unfused: (c*x - s*y, s*x + c*y)
__device__ __forceinline__ void rot2(double c, double s, double x, double y, double& ox, double& oy) {
    ox = c * x - s * y;
    oy = s * x + c * y;
}

__device__ __forceinline__ double horner4(const double* p, double x) {   // np.polyval, degree 4
    double y = 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k) y = y * x + p[k];
    return y;
}

// sin and cos for |x| <= ~2*pi (angles drawn in [-180, 180) degrees): two-term Cody-Waite reduction
// by pi/2 and the fdlibm kernel polynomials; <= 1.3 ulp (checked against mpmath on the host), about a
// third of the instructions of the general-range library pair.  Accuracy-only code: the oracle calls
// numpy's cos/sin, so operation order here is free.
__device__ __forceinline__ void sincos_small(double x, double& sn, double& cs) {
    const double k = rint(x * 0.6366197723675814);
    const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    const double z = r * r;
    const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                      z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double s = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double c = 1.0 - (0.5 * z - (z * z) * pc);
    const int q = ((int)k) & 3;
    sn = q == 0 ? s : q == 1 ? c : q == 2 ? -s : -c;
    cs = q == 0 ? c : q == 1 ? -s : q == 2 ? -c : s;
}

// Corridor resample = torchvision 0.12's TENSOR path (functional.rotate / functional.affine -> _gen_affine_grid ->
// grid_sample(nearest, zeros, align_corners=False); Path.py:160-161,175, MapGenerate.py:103-106), in the float32 operations of
// the torch primitives it calls — stated and pinned in oracle/edage_np.py ("rasters": affine_source_index; the CPU test
// test_resample_rule_is_torchvisions_tensor_path_* compares that statement with torch's own linspace / bmm / grid_sample).
// One axis of the map: (a, b, c) = a row of the float32 inverse matrix divided by half the image size, hs = size / 2;
// source index of the output pixel whose centre is (x, y) = (j + 0.5 - w/2, i + 0.5 - h/2):
//   g = fl(fma(y, b, fl(x a)) + c);  u = fl(fl((g + 1) hs) - 0.5);  index = nearbyint(u).
struct TvAxis { float a, b, c, hs; };
constexpr double PPN_DEG2RAD = 3.141592653589793 / 180.0;              // math.radians: x * (pi / 180)
__device__ __forceinline__ TvAxis tv_axis(double ma, double mb, double mc, int size) {
    const float den = 0.5f * (float)size;
    TvAxis t;
    t.a = __fdiv_rn((float)ma, den); t.b = __fdiv_rn((float)mb, den); t.c = __fdiv_rn((float)mc, den);
    t.hs = (float)size * 0.5f;
    return t;
}
__device__ __forceinline__ int tv_src(const TvAxis& t, float x, float y) {
    const float g = __fadd_rn(__fmaf_rn(y, t.b, __fmul_rn(x, t.a)), t.c);
    const float u = __fsub_rn(__fmul_rn(__fadd_rn(g, 1.0f), t.hs), 0.5f);
    return (int)rintf(u);
}

// Geometry of the reference's obstacle raster (Path.plot_obstacles, Path.py:36-49): see oracle/edage_np.py raster_geometry for
// the derivation.  Output pixel (i, j) shows the data point X = (j + 0.5) * ax + bx, Y = (i + 0.5) * ay + by and a circle of radius r
// inks the ellipse with semi-axes r + sx, r + sy around its centre; carried into the pixel-centre frame that is the axis-aligned
// ellipse of raster_ellipse().  The same double operations in the same order as the oracle (1 / ax and 1 / ay as constants).
struct RasterGeom { double bx, by, sx, sy; };
constexpr double PPN_RASTER_IAX = 446.4 / 444.0, PPN_RASTER_IAY = 332.64 / 330.0;
__device__ __forceinline__ RasterGeom raster_geom(int R) {
    const double Rd = (double)R;
    RasterGeom g;
    g.bx = (73.0 - 72.0) / 446.4 * Rd; g.by = (53.0 - 51.84) / 332.64 * Rd;
    g.sx = 0.625 / 446.4 * Rd; g.sy = 0.625 / 332.64 * Rd;
    return g;
}
// centre (col, row), semi-axes and (ex * ey)^2 of a circle's raster in pixel-centre coordinates
struct RasterEllipse { double cxp, cyp, ex, ey, rhs; };
__device__ __forceinline__ RasterEllipse raster_ellipse(double cx, double cy, double r, const RasterGeom& g) {
    RasterEllipse e;
    e.cxp = (cx - g.bx) * PPN_RASTER_IAX; e.cyp = (cy - g.by) * PPN_RASTER_IAY;
    e.ex = (r + g.sx) * PPN_RASTER_IAX; e.ey = (r + g.sy) * PPN_RASTER_IAY;
    const double q = e.ex * e.ey;
    e.rhs = q * q;
    return e;
}
// exact predicate of the raster rule for pixel column j: b2 = (((i + 0.5) - cyp) * ex)^2 of the row
__device__ __forceinline__ bool disc_pred(int j, double cxp, double ey, double b2, double rhs) {
    const double a = (((double)j + 0.5) - cxp) * ey;
    return a * a + b2 <= rhs;
}
// How far beyond its radius r — measured from its centre in pixel-centre coordinates — the raster of ANY circle can reach inside an
// R x R image: a pixel centre (xp, yp) in [0, R]^2 shows the data point (xp * ax + bx, yp * ay + by), at most (Dx, Dy) away from it,
// and that point lies within r + max(sx, sy) of the circle's centre.  An upper bound (it only feeds the decision to skip a pass
// that would be a no-op); 1.88 px at R = 256.
__device__ __forceinline__ double raster_reach_max(int R) {
    const RasterGeom g = raster_geom(R);
    const double Rd = (double)R;
    const double dx = fmax(fabs(g.bx), fabs(g.bx - Rd * (1.0 - 1.0 / PPN_RASTER_IAX)));
    const double dy = fmax(fabs(g.by), fabs(g.by - Rd * (1.0 - 1.0 / PPN_RASTER_IAY)));
    return sqrt(dx * dx + dy * dy) + fmax(g.sx, g.sy) + 1e-3;
}

__device__ __forceinline__ double dist2d(double ax, double ay, double bx, double by) {
    double dx = ax - bx, dy = ay - by;
    return sqrt(dx * dx + dy * dy);
}

}  // namespace ppn
