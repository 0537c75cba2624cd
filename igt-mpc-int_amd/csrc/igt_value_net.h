// igt_value_net.h -- terminal value network of the gt_mpc cost (config 5).
//
// Reference: mpc.py:367-369  J -= V(Wn (x_N - mu_f)) * sigma_t + mu_t, with
//   x_N = [s_tv, v_tv, e_tv, s_N - s_tv, v_N - v_tv, e_ego - e_tv]        (mpc.py:326-338)
//   V   = Linear(6,128)-tanh-Linear(128,128)-tanh-[Linear(128,128)-tanh-]Linear(128,1)   (model.py:14-51)
//
// Only s_N and v_N differ between the candidates of a scenario, so the first layer is affine in two
// scalars:  a1 = p(b) + q s_N + r v_N  with  A1 = W1 Wn,  p(b) = b1 - A1 mu_f + A1 f0(b),  q = A1[:,3],
// r = A1[:,4]  (value_prep_kernel computes p once per scenario).  The hidden layers are evaluated per
// candidate with one lane per candidate: activations of the lane live in its own LDS column (no
// barriers), weights are wave-uniform and arrive through scalar loads as [i][j]-transposed rows, and 16
// independent accumulators per lane keep the FMA stream issue-bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace igt {

constexpr int VN_H = 128;      // hidden width of every shipped checkpoint (sc*_config.yaml: hidden_size 128)

template <typename T>
struct DevNet {                // device pointers, all wave-uniform
    const T* A1;               // [128, 6]   W1 Wn
    const T* c1;               // [128]      b1 - A1 mu_f
    const T* WT[2];            // [128(i), 128(j)] transposed hidden->hidden weights
    const T* bias[2];          // [128]
    const T* wout;             // [128]
    T bout, sigma_t, mu_t;
    int n_hidden_mats;         // 1 (two hidden layers) or 2 (three hidden layers)
    const float* frag;         // f32 MFMA fragment block (float net only; see value_mfma_kernel)
    const double* fragd;       // f64 MFMA fragment block (double net only; see value_mfma_f64_kernel)
};

// MFMA fragment block of the float net (layout: value_mfma_kernel)
constexpr int FRAG_A1 = 4 * 4 * 64, FRAG_W = 4 * 64 * 64, FRAG_B = 4 * 16 * 2;
__host__ __device__ constexpr int frag_floats(int nm) { return FRAG_A1 + nm * FRAG_W + nm * FRAG_B + FRAG_B; }
__host__ __device__ constexpr int frag_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }   // C/D row of register r

// f64 MFMA fragment block of the double net (layout: value_mfma_f64_kernel), in doubles
constexpr int FRAGD_A1 = 8 * 2 * 64, FRAGD_W = 8 * 8 * 4 * 64, FRAGD_B = 8 * 4 * 4;
__host__ __device__ constexpr int fragd_doubles(int nm) { return FRAGD_A1 + nm * FRAGD_W + nm * FRAGD_B + FRAGD_B; }

template <typename T> __device__ __forceinline__ T tanh_t(T x);
template <> __device__ __forceinline__ float tanh_t<float>(float x) { return tanhf(x); }
template <> __device__ __forceinline__ double tanh_t<double>(double x) { return tanh(x); }

// p[b, i] = c1[i] + sum_k A1[i,k] f0[b,k],  f0 = [s_tv, v_tv, e_tv, -s_tv, -v_tv, e_ego - e_tv]
template <typename T>
__global__ __launch_bounds__(128) void value_prep_kernel(int B, DevNet<T> net, const T* __restrict__ tv_sv,
                                                         const T* __restrict__ enc, T* __restrict__ p_vec) {
    const int b = blockIdx.x, i = threadIdx.x;
    if (b >= B) return;
    const T s_tv = tv_sv[(size_t)b * 2 + 0], v_tv = tv_sv[(size_t)b * 2 + 1];
    const T e_ego = enc[(size_t)b * 2 + 0], e_tv = enc[(size_t)b * 2 + 1];
    const T f0[6] = {s_tv, v_tv, e_tv, -s_tv, -v_tv, e_ego - e_tv};
    T acc = net.c1[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) acc += net.A1[i * 6 + k] * f0[k];
    p_vec[(size_t)b * VN_H + i] = acc;
}

// one wave = 64 candidates of one scenario; one lane = one candidate.
// Writes the chunk's best (J, c) to part_J/part_c (ties -> lowest index) and, when cost_all != nullptr
// (rollout-all debug path), every candidate's total cost.
template <typename T>
__global__ __launch_bounds__(64) void value_kernel(int B, int C, DevNet<T> net, const T* __restrict__ p_vec,
                                                   const T* __restrict__ rec_sN, const T* __restrict__ rec_vN,
                                                   const double* __restrict__ rec_J,
                                                   const uint32_t* __restrict__ rec_viol,
                                                   double* __restrict__ part_J, int32_t* __restrict__ part_c,
                                                   T* __restrict__ cost_all, uint32_t* __restrict__ viol_all) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* const hA = reinterpret_cast<T*>(smem);
    T* const hB = hA + VN_H * 64;
    const int chunks = C / 64;
    const int gw = blockIdx.x;
    if (gw >= B * chunks) return;
    const int b = gw / chunks, chunk = gw - b * chunks;
    const int lane = threadIdx.x;
    const int c = chunk * 64 + lane;
    const size_t idx = (size_t)b * C + c;
    const T sN = rec_sN[idx], vN = rec_vN[idx];
    const T* __restrict__ p = p_vec + (size_t)b * VN_H;
    // layer 1 (affine in s_N, v_N) + tanh
    for (int i = 0; i < VN_H; ++i)
        hA[i * 64 + lane] = tanh_t<T>(p[i] + net.A1[i * 6 + 3] * sN + net.A1[i * 6 + 4] * vN);
    // hidden -> hidden layers, 16 output neurons at a time
    T V = net.bout;
    for (int m = 0; m < net.n_hidden_mats; ++m) {
        const T* __restrict__ src = (m & 1) ? hB : hA;
        T* __restrict__ dst = (m & 1) ? hA : hB;
        const T* __restrict__ WT = net.WT[m];
        const T* __restrict__ bias = net.bias[m];
        const bool last = (m == net.n_hidden_mats - 1);
        for (int t = 0; t < VN_H / 16; ++t) {
            T acc[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) acc[jj] = bias[t * 16 + jj];
            for (int i = 0; i < VN_H; ++i) {
                const T hv = src[i * 64 + lane];
                const T* __restrict__ w = WT + (size_t)i * VN_H + t * 16;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) acc[jj] += w[jj] * hv;
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const T hj = tanh_t<T>(acc[jj]);
                if (last) V += net.wout[t * 16 + jj] * hj;
                else dst[(t * 16 + jj) * 64 + lane] = hj;
            }
        }
    }
    // mpc.py:369: J -= V * sigma_t + mu_t
    const double Jt = rec_J[idx] - ((double)V * (double)net.sigma_t + (double)net.mu_t);
    unsigned viol = rec_viol[idx];
    const bool fin = fabs(Jt) < 1.79e308;
    if (cost_all) {
        cost_all[idx] = (T)Jt;
        viol_all[idx] = fin ? viol : (viol | 64u);
    }
    double bestJ = Jt;
    int bestC = (viol == 0 && fin) ? c : -1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double oJ = __shfl_xor(bestJ, off, 64);
        const int oC = __shfl_xor(bestC, off, 64);
        const bool take = (oC >= 0) && (bestC < 0 || oJ < bestJ || (oJ == bestJ && oC < bestC));
        if (take) { bestJ = oJ; bestC = oC; }
    }
    if (lane == 0 && part_J) { part_J[gw] = bestJ; part_c[gw] = bestC; }
}

// ---------------------------------------------------------------------------------------
// float specialisation, issue-bound form: the lane's 128 activations live in VGPRs (the i-loop is fully
// unrolled so every index is static), 16 accumulators per 16-neuron tile, weights by scalar loads that the
// unrolled body lets the compiler keep several rows ahead; tanh = 1 - 2/(exp(2x)+1) on the hardware
// exp2/rcp (abs error ~2e-7).  A 64-candidate chunk without a single feasible lane is skipped.
// ---------------------------------------------------------------------------------------
#ifdef IGT_KERNELS_TU   // non-template kernels: compiled by igt_kernels.hip only
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);        // exp(2x)
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

__device__ __forceinline__ void value_finish(int gw, int c, size_t idx, float V, unsigned viol, const DevNet<float>& net,
                                             const double* __restrict__ rec_J, double* __restrict__ part_J,
                                             int32_t* __restrict__ part_c, float* __restrict__ cost_all,
                                             uint32_t* __restrict__ viol_all) {
    const double Jt = rec_J[idx] - ((double)V * (double)net.sigma_t + (double)net.mu_t);   // mpc.py:369
    const bool fin = fabs(Jt) < 1.79e308;
    if (cost_all) {
        cost_all[idx] = (float)Jt;
        viol_all[idx] = fin ? viol : (viol | 64u);
    }
    double bestJ = Jt;
    int bestC = (viol == 0 && fin) ? c : -1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double oJ = __shfl_xor(bestJ, off, 64);
        const int oC = __shfl_xor(bestC, off, 64);
        const bool take = (oC >= 0) && (bestC < 0 || oJ < bestJ || (oJ == bestJ && oC < bestC));
        if (take) { bestJ = oJ; bestC = oC; }
    }
    if ((threadIdx.x & 63) == 0 && part_J) { part_J[gw] = bestJ; part_c[gw] = bestC; }
}

// two hidden layers (one 128x128 matrix): h1[128] in VGPRs, i-loop fully unrolled
__global__ __launch_bounds__(64) void value_kernel_f32_h2(int B, int C, DevNet<float> net, const float* __restrict__ p_vec,
                                                          const float* __restrict__ rec_sN,
                                                          const float* __restrict__ rec_vN,
                                                          const double* __restrict__ rec_J,
                                                          const uint32_t* __restrict__ rec_viol,
                                                          double* __restrict__ part_J, int32_t* __restrict__ part_c,
                                                          float* __restrict__ cost_all, uint32_t* __restrict__ viol_all) {
    const int chunks = C / 64;
    const int gw = blockIdx.x;
    if (gw >= B * chunks) return;
    const int b = gw / chunks, chunk = gw - b * chunks;
    const int lane = threadIdx.x;
    const int c = chunk * 64 + lane;
    const size_t idx = (size_t)b * C + c;
    const unsigned viol = rec_viol[idx];
    if (!cost_all && !__any(viol == 0)) {                    // nothing in this chunk can win
        if (lane == 0) { part_J[gw] = 0.0; part_c[gw] = -1; }
        return;
    }
    const float sN = rec_sN[idx], vN = rec_vN[idx];
    const float* __restrict__ p = p_vec + (size_t)b * VN_H;
    float h[VN_H];
#pragma unroll
    for (int i = 0; i < VN_H; ++i)
        h[i] = tanh_fast(fmaf(net.A1[i * 6 + 4], vN, fmaf(net.A1[i * 6 + 3], sN, p[i])));
    float V = net.bout;
    const float* __restrict__ WT = net.WT[0];
    const float* __restrict__ bias = net.bias[0];
    for (int t = 0; t < VN_H / 16; ++t) {
        float acc[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) acc[jj] = bias[t * 16 + jj];
#pragma unroll
        for (int i = 0; i < VN_H; ++i) {
            const float* __restrict__ w = WT + (size_t)i * VN_H + t * 16;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) acc[jj] = fmaf(w[jj], h[i], acc[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) V = fmaf(net.wout[t * 16 + jj], tanh_fast(acc[jj]), V);
    }
    value_finish(gw, c, idx, V, viol, net, rec_J, part_J, part_c, cost_all, viol_all);
}

// three hidden layers (two 128x128 matrices), no LDS: the third layer's 128 pre-activations are the
// register-resident accumulators; each finished h2 value is scattered into all of them at once, and h1 is
// recomputed per 16-neuron tile (7 instructions per use) instead of being stored.
__global__ __launch_bounds__(64) void value_kernel_f32_h3(int B, int C, DevNet<float> net, const float* __restrict__ p_vec,
                                                          const float* __restrict__ rec_sN,
                                                          const float* __restrict__ rec_vN,
                                                          const double* __restrict__ rec_J,
                                                          const uint32_t* __restrict__ rec_viol,
                                                          double* __restrict__ part_J, int32_t* __restrict__ part_c,
                                                          float* __restrict__ cost_all, uint32_t* __restrict__ viol_all) {
    const int chunks = C / 64;
    const int gw = blockIdx.x;
    if (gw >= B * chunks) return;
    const int b = gw / chunks, chunk = gw - b * chunks;
    const int lane = threadIdx.x;
    const int c = chunk * 64 + lane;
    const size_t idx = (size_t)b * C + c;
    const unsigned viol = rec_viol[idx];
    if (!cost_all && !__any(viol == 0)) {
        if (lane == 0) { part_J[gw] = 0.0; part_c[gw] = -1; }
        return;
    }
    const float sN = rec_sN[idx], vN = rec_vN[idx];
    const float* __restrict__ p = p_vec + (size_t)b * VN_H;
    const float* __restrict__ A1 = net.A1;
    const float* __restrict__ WT0 = net.WT[0];
    const float* __restrict__ WT1 = net.WT[1];
    float acc3[VN_H];
#pragma unroll
    for (int j = 0; j < VN_H; ++j) acc3[j] = net.bias[1][j];
    for (int t = 0; t < VN_H / 16; ++t) {
        float acc2[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) acc2[jj] = net.bias[0][t * 16 + jj];
#pragma unroll 4
        for (int i = 0; i < VN_H; ++i) {
            const float h1 = tanh_fast(fmaf(A1[i * 6 + 4], vN, fmaf(A1[i * 6 + 3], sN, p[i])));
            const float* __restrict__ w = WT0 + (size_t)i * VN_H + t * 16;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) acc2[jj] = fmaf(w[jj], h1, acc2[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const float h2 = tanh_fast(acc2[jj]);
            const float* __restrict__ w3 = WT1 + (size_t)(t * 16 + jj) * VN_H;
#pragma unroll
            for (int j = 0; j < VN_H; ++j) acc3[j] = fmaf(w3[j], h2, acc3[j]);
        }
    }
    float V = net.bout;
#pragma unroll
    for (int j = 0; j < VN_H; ++j) V = fmaf(net.wout[j], tanh_fast(acc3[j]), V);
    value_finish(gw, c, idx, V, viol, net, rec_J, part_J, part_c, cost_all, viol_all);
}

// ---------------------------------------------------------------------------------------
// Compact form used by the float solve path: only ~6 % of the lattice candidates survive the verdicts, so
// the search pass appends the feasible ones to a list (one wave-aggregated atomic per wave; a wave's entries
// are contiguous and belong to one scenario) and the value kernels walk that list with a grid-stride loop.
// Layer 1 is evaluated from the six features directly (6 FMAs per neuron with scalar weights), the
// scenario's winner is kept with one 64-bit atomicMin on (orderable float cost, candidate index).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long pack_key(float J, int c) {
    unsigned u = __float_as_uint(J);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // ascending order as unsigned
    return ((unsigned long long)u << 32) | (unsigned)c;
}
__device__ __forceinline__ float key_cost(unsigned long long key) {
    unsigned u = (unsigned)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

struct CompactRecs {           // feasible candidates appended by the search pass
    const unsigned* count;
    const int32_t* b;
    const int32_t* c;
    const float* sN;
    const float* vN;
    const double* J;           // cost without the terminal term
};

__device__ __forceinline__ void compact_features(const CompactRecs& R, unsigned e, bool live, const float* __restrict__ tv_sv,
                                                 const float* __restrict__ enc, float (&g)[6], int& b, int& c, double& J) {
    b = live ? R.b[e] : 0;
    c = live ? R.c[e] : 0;
    J = live ? R.J[e] : 0.0;
    const float sN = live ? R.sN[e] : 0.0f, vN = live ? R.vN[e] : 0.0f;
    const float s_tv = tv_sv[(size_t)b * 2 + 0], v_tv = tv_sv[(size_t)b * 2 + 1];
    const float e_ego = enc[(size_t)b * 2 + 0], e_tv = enc[(size_t)b * 2 + 1];
    // x_N = [s_tv, v_tv, e_tv, s_N - s_tv, v_N - v_tv, e_ego - e_tv]   (mpc.py:326-338)
    g[0] = s_tv; g[1] = v_tv; g[2] = e_tv; g[3] = sN - s_tv; g[4] = vN - v_tv; g[5] = e_ego - e_tv;
}

__device__ __forceinline__ float layer1(const DevNet<float>& net, int i, const float (&g)[6]) {
    float a = net.c1[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) a = fmaf(net.A1[i * 6 + k], g[k], a);
    return tanh_fast(a);
}

__device__ __forceinline__ void compact_finish(const DevNet<float>& net, bool live, int b, int c, double J, float V,
                                               unsigned long long* __restrict__ best_key) {
    const double Jt = J - ((double)V * (double)net.sigma_t + (double)net.mu_t);   // mpc.py:369
    const float Jf = (float)Jt;
    if (live && fabsf(Jf) < 3.0e38f) atomicMin(&best_key[b], pack_key(Jf, c));
}

// ---------------------------------------------------------------------------------------
// The network over the compact list on the matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate).
// This part of the path IS a dense contraction: [128 x 128] x [128 x 10^6 feasible candidates] per hidden layer.
// Everything is kept transposed, X = [neuron x candidate]: a layer is Y = W X, summed over X's ROW index, so a
// layer's accumulator tiles (column = candidate on the lane, 16 rows in the registers) are the next layer's B
// operands as they stand -- no LDS round trip, no shuffles.  The k order this implies (k-step (ti, r) pairs row
// 32 ti + (r&3) + 8 (r>>2) of lane half 0 with the same + 4 of lane half 1) is baked into the A fragments, which
// igt_set_value_net lays out once (frag block below) and every workgroup stages into LDS (68 / 134 KB).
//   wave = 32 candidates (lanes l and l+32 hold the same candidate, different rows)
//   layer 1:  [128 x 8] x [8 x 32], features (f0..f5, 1, 0): the bias rides as the 7th feature      16 MFMA
//   hidden :  4 output tiles x 64 k-steps, accumulators start at the bias fragment                  256 MFMA each
//   output :  64 FMAs per lane + one cross-half add
// frag block (floats): A1F[4][4][64] | WF[m][4][64][64] (m < n_hidden_mats) | BF[m][4][16][2] | WOF[4][16][2]
// ---------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM>
__global__ __launch_bounds__(512) void value_mfma_kernel(DevNet<float> net, CompactRecs R, const float* __restrict__ tv_sv,
                                                         const float* __restrict__ enc,
                                                         unsigned long long* __restrict__ best_key) {
    extern __shared__ float lds[];
    constexpr int NF = frag_floats(NM);
    for (int i = threadIdx.x; i < NF; i += 512) lds[i] = net.frag[i];
    __syncthreads();
    const float* A1F = lds;
    const float* WF = lds + FRAG_A1;
    const float* BF = WF + NM * FRAG_W;
    const float* WOF = BF + NM * FRAG_B;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const unsigned count = *R.count;
    const unsigned wave = blockIdx.x * 8u + (threadIdx.x >> 6), nwaves = gridDim.x * 8u;
    for (unsigned base = wave * 32u; base < count; base += nwaves * 32u) {
        const unsigned e = base + (unsigned)(lane & 31);
        const bool live = e < count;
        float g[6];
        int b, c;
        double J;
        compact_features(R, e, live, tv_sv, enc, g, b, c, J);
        // B operand of layer 1: feature k = 2 s + half
        const float xs[4] = {half ? g[1] : g[0], half ? g[3] : g[2], half ? g[5] : g[4], half ? 0.0f : 1.0f};
        f32x16 Ha[4], Hb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A1F[(t * 4 + s4) * 64 + lane], xs[s4], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) Ha[t][r] = tanh_fast(acc[r]);
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const float* W = WF + m * FRAG_W;
            const float* Bm = BF + m * FRAG_B;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = Bm[(t * 16 + r) * 2 + half];
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    // 16 A fragments at a time: without the fences the ILP scheduler hoists all 256 LDS reads of a
                    // layer and spills
                    float afr[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) afr[r] = W[(t * 64 + ti * 16 + r) * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float bop = (m == 0) ? Ha[ti][r] : Hb[ti][r];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[r], bop, acc, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (m == 0) Hb[t][r] = tanh_fast(acc[r]); else Ha[t][r] = tanh_fast(acc[r]);
                }
            }
        }
        float v = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float hv = (NM == 1) ? Hb[t][r] : Ha[t][r];
                v = fmaf(WOF[(t * 16 + r) * 2 + half], hv, v);
            }
        v += __shfl_xor(v, 32, 64);
        compact_finish(net, live && half == 0, b, c, J, v + net.bout, best_key);
    }
}

// ---------------------------------------------------------------------------------------
// The double net over the compact list on the matrix cores: v_mfma_f64_16x16x4_f64 (f64 in, f64 accumulate).
// Same transposed formulation as the float kernel, X = [neuron x candidate], a layer is Y = W X.  With this
// instruction the chain needs no re-ordering at all: a lane (g = lane >> 4, col = lane & 15) holds rows g + 4 r of a
// 16 x 16 output tile in its 4 result registers, and as B operand of a k-step it must supply row k = g -- so result
// register r of tile T IS the B operand of the k-step that covers neurons 16 T + 4 r + (0..3), as it stands.
//   wave = 16 candidates (the 4 lanes of a column hold the same candidate, different rows)
//   layer 1:  [128 x 8] x [8 x 16], features (f0..f5, 1, 0): the bias rides as the 7th feature     16 MFMA
//   hidden :  8 output tiles x 32 k-steps, accumulators start at the bias fragment                  256 MFMA each
//   output :  32 FMAs per lane + two cross-lane adds
// fragd block (doubles): A1F[8][2][64] | WF[m][8][8][4][64] | BF[m][8][4][4] | WOF[8][4][4]; the first hidden matrix
// (128 KB) is staged in LDS, a second one is read through L2.  tanh = 1 - 2/(exp(2x)+1) with a double exp
// (Cody-Waite reduction, degree-12 polynomial, v_ldexp_f64): absolute error ~2e-16.
// ---------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double tanh_d(double x) {
    const double y = fmin(fmax(2.0 * x, -80.0), 80.0);
    const double n = __builtin_rint(y * 1.4426950408889634);
    double r = fma(-n, 6.93147180369123816490e-01, y);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = fma(r, 1.0 / 479001600.0, 1.0 / 39916800.0);
    p = fma(r, p, 1.0 / 3628800.0);
    p = fma(r, p, 1.0 / 362880.0);
    p = fma(r, p, 1.0 / 40320.0);
    p = fma(r, p, 1.0 / 5040.0);
    p = fma(r, p, 1.0 / 720.0);
    p = fma(r, p, 1.0 / 120.0);
    p = fma(r, p, 1.0 / 24.0);
    p = fma(r, p, 1.0 / 6.0);
    p = fma(r, p, 0.5);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    const double e = ldexp(p, (int)n) + 1.0;                  // exp(2x) + 1
    double q = __builtin_amdgcn_rcp(e);
    q = fma(fma(-e, q, 1.0), q, q);
    q = fma(fma(-e, q, 1.0), q, q);
    return fma(-2.0, q, 1.0);
}

// LDS image of the fragment block: the first hidden matrix, then the small fragments (layer 1, biases, output row)
constexpr int FRAGD_LDS_SMALL = FRAGD_A1 + 2 * FRAGD_B + FRAGD_B;      // A1F | BF[0..1] | WOF
constexpr int FRAGD_LDS = FRAGD_W + FRAGD_LDS_SMALL;                   // doubles (139 KB)

// live_idx[0 .. *live_count): the entries to evaluate (value_select_kernel); the entries themselves sit in their units' slots.
template <int NM>
__global__ __launch_bounds__(512) void value_mfma_f64_kernel(DevNet<double> net, const unsigned* __restrict__ live_count,
                                                             const unsigned* __restrict__ live_idx,
                                                             const int32_t* __restrict__ rec_b,
                                                             const double* __restrict__ rec_sN,
                                                             const double* __restrict__ rec_vN, double* __restrict__ rec_J,
                                                             const double* __restrict__ tv_sv, const double* __restrict__ enc) {
    extern __shared__ double ldsd[];
    const double* __restrict__ WFg = net.fragd + FRAGD_A1;
    double* const A1F = ldsd + FRAGD_W;
    double* const BF = A1F + FRAGD_A1;
    double* const WOF = BF + 2 * FRAGD_B;
    for (int i = threadIdx.x; i < FRAGD_W; i += 512) ldsd[i] = WFg[i];
    for (int i = threadIdx.x; i < FRAGD_A1; i += 512) A1F[i] = net.fragd[i];
    for (int i = threadIdx.x; i < NM * FRAGD_B; i += 512) BF[i] = WFg[NM * FRAGD_W + i];
    for (int i = threadIdx.x; i < FRAGD_B; i += 512) WOF[i] = WFg[NM * FRAGD_W + NM * FRAGD_B + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
    const unsigned count = *live_count;
    const unsigned wave = blockIdx.x * 8u + (threadIdx.x >> 6), nwaves = gridDim.x * 8u;
    for (unsigned base = wave * 16u; base < count; base += nwaves * 16u) {
        const unsigned slot = base + (unsigned)col;
        const bool live = slot < count;
        const unsigned e = live ? live_idx[slot] : 0u;
        const int b = live ? rec_b[e] : 0;
        const double sN = live ? rec_sN[e] : 0.0, vN = live ? rec_vN[e] : 0.0;
        const double s_tv = tv_sv[(size_t)b * 2 + 0], v_tv = tv_sv[(size_t)b * 2 + 1];
        const double e_ego = enc[(size_t)b * 2 + 0], e_tv = enc[(size_t)b * 2 + 1];
        // x_N = [s_tv, v_tv, e_tv, s_N - s_tv, v_N - v_tv, e_ego - e_tv]   (mpc.py:326-338); B operand: feature 4 s + g
        const double x0 = g == 0 ? s_tv : g == 1 ? v_tv : g == 2 ? e_tv : sN - s_tv;
        const double x1 = g == 0 ? vN - v_tv : g == 1 ? e_ego - e_tv : g == 2 ? 1.0 : 0.0;
        // the second matrix is read through L2 at compile-time offsets from this pointer; it is made opaque per
        // iteration so that the 256 fragment addresses are formed where they are used, not hoisted out of the loop
        // (kept live they would spill ~600 registers)
        const double* wg = WFg + lane;
        asm volatile("" : "+v"(wg));
        f64x4 Ha[8], Hb[8];
        // Two output tiles at a time: a dependent v_mfma_f64_16x16x4_f64 chain issues every ~214 cycles on gfx950, two
        // interleaved chains every ~160 (tools/mfma64_microbench.hip), and a wave of the other SIMD slot fills the rest.
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A1F[(t * 2 + 0) * 64 + lane], x0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A1F[(t * 2 + 2) * 64 + lane], x0, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A1F[(t * 2 + 1) * 64 + lane], x1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A1F[(t * 2 + 3) * 64 + lane], x1, acc1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) { Ha[t][r] = tanh_d(acc0[r]); Ha[t + 1][r] = tanh_d(acc1[r]); }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) {
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                f64x4 acc0, acc1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc0[r] = BF[m * FRAGD_B + (t * 4 + r) * 4 + g];
                    acc1[r] = BF[m * FRAGD_B + ((t + 1) * 4 + r) * 4 + g];
                }
#pragma unroll
                for (int ti = 0; ti < 8; ++ti) {
                    // 8 A fragments at a time: without the fences the ILP scheduler hoists a whole layer's fragment reads
                    // and spills hundreds of registers
                    double af0[4], af1[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i0 = ((t * 8 + ti) * 4 + r) * 64, i1 = (((t + 1) * 8 + ti) * 4 + r) * 64;
                        af0[r] = (m == 0) ? ldsd[i0 + lane] : wg[(size_t)m * FRAGD_W + i0];
                        af1[r] = (m == 0) ? ldsd[i1 + lane] : wg[(size_t)m * FRAGD_W + i1];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double bop = (m == 0) ? Ha[ti][r] : Hb[ti][r];
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af0[r], bop, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af1[r], bop, acc1, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (m == 0) { Hb[t][r] = tanh_d(acc0[r]); Hb[t + 1][r] = tanh_d(acc1[r]); }
                    else { Ha[t][r] = tanh_d(acc0[r]); Ha[t + 1][r] = tanh_d(acc1[r]); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        double v = 0.0;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double hv = (NM == 1) ? Hb[t][r] : Ha[t][r];
                v = fma(WOF[(t * 4 + r) * 4 + g], hv, v);
            }
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (live && g == 0) rec_J[e] = rec_J[e] - ((v + net.bout) * net.sigma_t + net.mu_t);      // mpc.py:369
    }
}

// ---------------------------------------------------------------------------------------
// Pruning the compact list before the network runs (double path).  The cost of an entry is J_e - V(x_e) (mpc.py:369) and
// the hidden layers are tanh, so V is bounded on any box of features.  Per scenario -- only s_N and v_N differ between its
// entries -- one wave
//   * takes the box [s_N] x [v_N] of the scenario's entries and the entry a with the smallest J,
//   * pushes the box through the network as centre / radius intervals (affine layers: c' = W c + b, r' = |W| r; tanh is
//     monotone: [tanh(c - r), tanh(c + r)]) -> V <= V_hi on the box, and evaluates V(x_a) exactly,
//   * writes thr[b] = (J_a - V(x_a)) + V_hi: an entry with J_e > thr[b] costs more than entry a whatever its V, it cannot
//     be the arg-min.  value_prune_kernel drops those (their cost becomes +inf, which unit_reduce_kernel skips) and
//     lists the others for value_mfma_f64_kernel.
// Nothing is approximated: the surviving entries get the same costs as without pruning, the winner is the same entry.
// Measured on the benchmark batch (tools/prune_probe.py, identity normalisation): 16 % of the tracking family's list
// survives with V_GT_sc1, 39 % with V_GT_sc3.  The bound costs about as much as 16 entries of a scenario, so a scenario
// with fewer than min_entries entries (the lattice's 14 on average) gets thr = +inf and keeps them all.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void value_bound_kernel(int B, int W, int min_entries, int nm, DevNet<double> net,
                                                         const int2* __restrict__ unit_seg,
                                                         const double* __restrict__ rec_sN,
                                                         const double* __restrict__ rec_vN,
                                                         const double* __restrict__ rec_J,
                                                         const double* __restrict__ tv_sv, const double* __restrict__ enc,
                                                         double* __restrict__ thr) {
    __shared__ double hc[2][VN_H], hr[2][VN_H], ha[2][VN_H];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    int n_total = 0;
    for (int p = 0; p < W; ++p) n_total += unit_seg[(size_t)b * W + p].y;
    if (n_total < min_entries) {                       // wave-uniform
        if (lane == 0) thr[b] = (double)INFINITY;
        return;
    }
    double smin = 1e300, smax = -1e300, vmin = 1e300, vmax = -1e300, Ja = 1e300, sa = 0.0, va = 0.0;
    for (int p = 0; p < W; ++p) {
        const int2 sg = unit_seg[(size_t)b * W + p];
        for (int e = sg.x + lane; e < sg.x + sg.y; e += 64) {
            const double s = rec_sN[e], v = rec_vN[e], J = rec_J[e];
            smin = fmin(smin, s); smax = fmax(smax, s); vmin = fmin(vmin, v); vmax = fmax(vmax, v);
            if (J < Ja) { Ja = J; sa = s; va = v; }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        smin = fmin(smin, __shfl_xor(smin, off, 64)); smax = fmax(smax, __shfl_xor(smax, off, 64));
        vmin = fmin(vmin, __shfl_xor(vmin, off, 64)); vmax = fmax(vmax, __shfl_xor(vmax, off, 64));
        const double oJ = __shfl_xor(Ja, off, 64), os = __shfl_xor(sa, off, 64), ov = __shfl_xor(va, off, 64);
        // any entry with the smallest J will do as the anchor; ties are broken on (s, v) only to keep the lanes in agreement
        if (oJ < Ja || (oJ == Ja && (os < sa || (os == sa && ov < va)))) { Ja = oJ; sa = os; va = ov; }
    }
    const double s_tv = tv_sv[(size_t)b * 2 + 0], v_tv = tv_sv[(size_t)b * 2 + 1];
    const double e_ego = enc[(size_t)b * 2 + 0], e_tv = enc[(size_t)b * 2 + 1];
    const double f0[6] = {s_tv, v_tv, e_tv, -s_tv, -v_tv, e_ego - e_tv};
    const double sc = 0.5 * (smin + smax), sr = 0.5 * (smax - smin) * (1.0 + 1e-12) + 1e-300;
    const double vc = 0.5 * (vmin + vmax), vr = 0.5 * (vmax - vmin) * (1.0 + 1e-12) + 1e-300;
    constexpr double TANH_EPS = 1e-15;                 // tanh_d's absolute error, with room
#pragma unroll
    for (int q = 0; q < 2; ++q) {                      // layer 1: affine in (s_N, v_N)   (value_prep_kernel's p, q, r)
        const int i = lane + 64 * q;
        double p = net.c1[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) p = fma(net.A1[i * 6 + k], f0[k], p);
        const double qi = net.A1[i * 6 + 3], ri = net.A1[i * 6 + 4];
        const double c = fma(ri, vc, fma(qi, sc, p)), r = fma(fabs(ri), vr, fabs(qi) * sr) + 1e-13 * (1.0 + fabs(c));
        const double lo = tanh_d(c - r) - TANH_EPS, hi = tanh_d(c + r) + TANH_EPS;
        hc[0][i] = 0.5 * (lo + hi); hr[0][i] = 0.5 * (hi - lo);
        ha[0][i] = tanh_d(fma(ri, va, fma(qi, sa, p)));
    }
    __syncthreads();
    int cur = 0;
    for (int m = 0; m < nm; ++m) {
        const double* __restrict__ WT = net.WT[m];
        double c0 = net.bias[m][lane], c1 = net.bias[m][lane + 64], r0 = 0.0, r1 = 0.0, a0 = c0, a1 = c1;
        for (int i = 0; i < VN_H; ++i) {
            const double w0 = WT[(size_t)i * VN_H + lane], w1 = WT[(size_t)i * VN_H + lane + 64];
            const double xc = hc[cur][i], xr = hr[cur][i], xa = ha[cur][i];
            c0 = fma(w0, xc, c0); r0 = fma(fabs(w0), xr, r0); a0 = fma(w0, xa, a0);
            c1 = fma(w1, xc, c1); r1 = fma(fabs(w1), xr, r1); a1 = fma(w1, xa, a1);
        }
        r0 += 1e-13 * (1.0 + fabs(c0)); r1 += 1e-13 * (1.0 + fabs(c1));      // rounding of the 128-term sums, with room
        const double l0 = tanh_d(c0 - r0) - TANH_EPS, h0 = tanh_d(c0 + r0) + TANH_EPS;
        const double l1 = tanh_d(c1 - r1) - TANH_EPS, h1 = tanh_d(c1 + r1) + TANH_EPS;
        hc[cur ^ 1][lane] = 0.5 * (l0 + h0); hr[cur ^ 1][lane] = 0.5 * (h0 - l0); ha[cur ^ 1][lane] = tanh_d(a0);
        hc[cur ^ 1][lane + 64] = 0.5 * (l1 + h1); hr[cur ^ 1][lane + 64] = 0.5 * (h1 - l1); ha[cur ^ 1][lane + 64] = tanh_d(a1);
        __syncthreads();
        cur ^= 1;
    }
    double cV = 0.0, rV = 0.0, aV = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        const double w = net.wout[j];
        cV = fma(w, hc[cur][j], cV); rV = fma(fabs(w), hr[cur][j], rV); aV = fma(w, ha[cur][j], aV);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        cV += __shfl_xor(cV, off, 64); rV += __shfl_xor(rV, off, 64); aV += __shfl_xor(aV, off, 64);
    }
    if (lane == 0) {
        // scaled value (mpc.py:369): V sigma_t + mu_t; upper end of the interval whatever the sign of sigma_t
        const double v_hi = fma(cV + net.bout, net.sigma_t, net.mu_t) + fabs(net.sigma_t) * rV;
        const double v_a = fma(aV + net.bout, net.sigma_t, net.mu_t);
        const double t = (Ja - v_a) + v_hi;
        thr[b] = t + 1e-9 * (1.0 + fabs(Ja) + fabs(v_a) + fabs(v_hi));     // evaluation order differs from the MFMA kernel's
    }
}

// The dense list the network runs over: every unit's entries with J_e <= thr[b] (the others cannot win: their cost becomes
// +inf), in any order.  Each wave owns a contiguous range of units, counts its keeps, reserves its share of live_idx with ONE
// atomicAdd and fills it -- a few thousand atomics per launch instead of one per unit.
__global__ __launch_bounds__(256) void value_select_kernel(int n_units, int W, const int2* __restrict__ unit_seg,
                                                           double* __restrict__ rec_J, const double* __restrict__ thr,
                                                           unsigned* __restrict__ live_count, unsigned* __restrict__ live_idx) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = gridDim.x * (blockDim.x >> 6);
    const int per = (n_units + nwaves - 1) / nwaves;
    const int u0 = wave * per, u1 = (u0 + per < n_units) ? u0 + per : n_units;
    // 64 units at a time: lane j loads unit j's segment and threshold (one coalesced load each), the wave then walks the
    // units with the segment broadcast from lane j; lane j keeps unit j's keep-mask for the second pass
    for (int t0 = u0; t0 < u1; t0 += 64) {
        const int nu = (u1 - t0 < 64) ? u1 - t0 : 64;
        const int2 mine = lane < nu ? unit_seg[t0 + lane] : make_int2(0, 0);
        const double my_thr = lane < nu ? thr[(t0 + lane) / W] : 0.0;
        unsigned long long my_mask = 0ull;
        unsigned total = 0;
        for (int j = 0; j < nu; ++j) {
            const int base = __shfl(mine.x, j, 64), n = __shfl(mine.y, j, 64);
            const double tj = __shfl(my_thr, j, 64);
            const bool keep = lane < n && rec_J[base + lane] <= tj;
            const unsigned long long m = __ballot(keep);
            if (lane == j) my_mask = m;
            total += (unsigned)__popcll(m);
        }
        unsigned out = 0;
        if (lane == 0 && total) out = atomicAdd(live_count, total);
        out = __builtin_amdgcn_readfirstlane(out);
        for (int j = 0; j < nu; ++j) {
            const int base = __shfl(mine.x, j, 64), n = __shfl(mine.y, j, 64);
            const unsigned long long m = ((unsigned long long)(unsigned)__shfl((int)(my_mask >> 32), j, 64) << 32) |
                                         (unsigned)__shfl((int)(my_mask & 0xffffffffull), j, 64);
            if ((m >> lane) & 1ull) live_idx[out + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)(base + lane);
            else if (lane < n) rec_J[base + lane] = (double)INFINITY;
            out += (unsigned)__popcll(m);
        }
    }
}

// a unit's entries of the compact list (contiguous: [base, base + n)) -> the unit's best (J, c)
__global__ __launch_bounds__(256) void unit_reduce_kernel(int n_units, const int2* __restrict__ unit_seg,
                                                          const double* __restrict__ rec_J, const int32_t* __restrict__ rec_c,
                                                          double* __restrict__ part_J, int32_t* __restrict__ part_c) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_units) return;
    const int2 sg = unit_seg[u];
    double J = 0.0;
    int c = -1;
    for (int e = sg.x; e < sg.x + sg.y; ++e) {
        const double Je = rec_J[e];
        const int ce = rec_c[e];
        if (fabs(Je) < 1.79e308 && (c < 0 || Je < J || (Je == J && ce < c))) { J = Je; c = ce; }
    }
    part_J[u] = J; part_c[u] = c;
}

// best_key[B] -> one partial per scenario (part_J, part_c with W = 1) for refine / emit
__global__ __launch_bounds__(256) void keys_to_partials_kernel(int B, const unsigned long long* __restrict__ best_key,
                                                               double* __restrict__ part_J, int32_t* __restrict__ part_c) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned long long k = best_key[b];
    const bool any = k != 0xffffffffffffffffull;
    part_c[b] = any ? (int32_t)(unsigned)(k & 0xffffffffull) : -1;
    part_J[b] = any ? (double)key_cost(k) : 0.0;
}

#endif  // IGT_KERNELS_TU

}  // namespace igt
