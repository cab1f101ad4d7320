// Fused parameter update: per-parameter L2 clip (utils.py:106-115) + the optimizer rule + teacher EMA
// (main_esvit.py:587-590) in two multi-tensor launches with no host synchronisation (the reference does one .item() per
// tensor).  Rules (main_esvit.py:408-415): AdamW (torch.optim.AdamW as driven by main_esvit.py:506-510,574), SGD with
// momentum (torch.optim.SGD(lr=0, momentum=0.9)), LARS (utils.py:519-557).  SGD / LARS keep their one state tensor
// (momentum_buffer / mu) in the exp_avg slot; LARS needs |p| and |clipped g + wd p| per tensor, which the statistics
// pass delivers as (sum g^2, sum p^2, sum g p).
//
// Tensor table (device, int64[ntensors * 12]):
//   0 p  1 g  2 exp_avg  3 exp_avg_sq  4 teacher_p  5 numel  6 group (0: weight decay, 1: none)
//   7 flags (bit0: has gradient this step)  8 bias corrections: bits(1-beta1^t) | bits(1-beta2^t) << 32
//   9 reserved  10 bf16 copy of p (0 = none)  11 bf16 copy of teacher_p (0 = none): the activation-dtype weights the
//   next forward's GEMMs read, written here instead of by ~120 separate cast launches per step
// Chunk table (device, int32[nchunks * 2]): [tensor id, chunk index]; a chunk is 4096 elements and
// is processed by one 256-thread workgroup with float4 accesses.  HBM-bound: 9 floats of traffic
// per parameter (read p,g,m,v,teacher; write p,m,v,teacher) + 1 for the two bf16 copies.
#include "common.h"
#include "../../include/esvit_hip.h"

namespace {

constexpr int CHUNK = 4096;
constexpr int TFIELDS = 12;

// STATS 1: sqnorms[t] = sum g^2;  STATS 3: sqnorms[3t..3t+2] = (sum g^2, sum p^2, sum g p)
template <int STATS>
__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const long* __restrict__ tensors, const int* __restrict__ chunks,
                                                          float* __restrict__ sqnorms) {
    __shared__ float scratch[4];
    const int tid = chunks[2 * blockIdx.x], ci = chunks[2 * blockIdx.x + 1];
    const long* tt = tensors + (long)tid * TFIELDS;
    if (!(tt[7] & 1)) return;
    const float* g = reinterpret_cast<const float*>(tt[1]);
    const float* p = reinterpret_cast<const float*>(tt[0]);
    const long n = tt[5];
    const long base = (long)ci * CHUNK;
    float s = 0.f, sp = 0.f, sgp = 0.f;
#pragma unroll
    for (int i = 0; i < CHUNK / 256 / 4; ++i) {
        const long o = base + (i * 256 + threadIdx.x) * 4;
        if (o + 4 <= n) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(g + o);
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            if constexpr (STATS == 3) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(p + o);
                sp += w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3];
                sgp += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
            }
        } else {
            for (long j = o; j < n; ++j) {
                s += g[j] * g[j];
                if constexpr (STATS == 3) {
                    sp += p[j] * p[j];
                    sgp += g[j] * p[j];
                }
            }
        }
    }
    s = block_sum<256>(s, scratch);
    if (threadIdx.x == 0) atomicAdd(sqnorms + (long)tid * STATS, s);
    if constexpr (STATS == 3) {
        __syncthreads();
        sp = block_sum<256>(sp, scratch);
        if (threadIdx.x == 0) atomicAdd(sqnorms + (long)tid * 3 + 1, sp);
        __syncthreads();
        sgp = block_sum<256>(sgp, scratch);
        if (threadIdx.x == 0) atomicAdd(sqnorms + (long)tid * 3 + 2, sgp);
    }
}

// Non-finite guard: the reference reads loss.item() every iteration and exits BEFORE the update when it is not finite
// (main_esvit.py:546-551).  There is no host synchronisation here, so the update kernels look at the per-tensor statistics
// themselves: if any of them is NaN / inf (a non-finite loss poisons every gradient), the whole update -- student, moments,
// teacher EMA, weight copies -- is skipped and the state stays what it was; the host finds the non-finite loss at its next look.
// `skipped` (optional device counter): bumped once per skipped update launch, so that the host can tell -- at its next look -- how many
// updates did not happen (a gradient overflow with a FINITE loss would otherwise go unnoticed) and take them back out of its step counts.
__device__ __forceinline__ bool stats_not_finite(const float* __restrict__ stats, int nstats, int* __restrict__ skipped) {
    int bad = 0;
    for (int i = threadIdx.x; i < nstats; i += 256) bad |= !(fabsf(stats[i]) <= 3.0e38f);
    const bool any = __syncthreads_or(bad) != 0;
    if (any && skipped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1);
    return any;
}

__device__ __forceinline__ void adamw_elem(float& p, float g, float& m, float& v, float decay, float b1, float b2, float step_size,
                                           float inv_sqrt_bc2, float eps) {
    p *= decay;
    m = m + (1.f - b1) * (g - m);
    v = b2 * v + (1.f - b2) * g * g;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void clip_adamw_ema_kernel(const long* __restrict__ tensors, const int* __restrict__ chunks,
                                                             const float* __restrict__ sqnorms, int nstats, float clip, float lr, float wd,
                                                             float b1, float b2, float eps, float ema_m, int* __restrict__ skipped) {
    if (stats_not_finite(sqnorms, nstats, skipped)) return;
    const int tid = chunks[2 * blockIdx.x], ci = chunks[2 * blockIdx.x + 1];
    const long* tt = tensors + (long)tid * TFIELDS;
    float* p = reinterpret_cast<float*>(tt[0]);
    const float* g = reinterpret_cast<const float*>(tt[1]);
    float* m = reinterpret_cast<float*>(tt[2]);
    float* v = reinterpret_cast<float*>(tt[3]);
    float* tp = reinterpret_cast<float*>(tt[4]);
    bf16* pb = reinterpret_cast<bf16*>(tt[10]);  // optional bf16 copies the next forward reads (student / teacher)
    bf16* tb = reinterpret_cast<bf16*>(tt[11]);
    const long n = tt[5];
    const bool has_grad = tt[7] & 1;
    const float decay = (tt[6] == 0) ? 1.f - lr * wd : 1.f;
    const float bc1 = __uint_as_float((unsigned)(tt[8] & 0xffffffffL));
    const float bc2 = __uint_as_float((unsigned)((unsigned long)tt[8] >> 32));
    float gscale = 1.f;
    if (has_grad && clip > 0.f) {
        const float coef = clip / (sqrtf(sqnorms[tid]) + 1e-6f);
        if (coef < 1.f) gscale = coef;
    }
    const float step_size = lr / bc1;
    const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    const long base = (long)ci * CHUNK;
#pragma unroll
    for (int i = 0; i < CHUNK / 256 / 4; ++i) {
        const long o = base + (i * 256 + threadIdx.x) * 4;
        if (o + 4 <= n) {
            f32x4 pv = *reinterpret_cast<f32x4*>(p + o);
            if (has_grad) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(g + o) * gscale;
                f32x4 mv = *reinterpret_cast<f32x4*>(m + o), vv = *reinterpret_cast<f32x4*>(v + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = pv[e], me = mv[e], ve = vv[e];
                    adamw_elem(pe, gv[e], me, ve, decay, b1, b2, step_size, inv_sqrt_bc2, eps);
                    pv[e] = pe;
                    mv[e] = me;
                    vv[e] = ve;
                }
                *reinterpret_cast<f32x4*>(p + o) = pv;
                *reinterpret_cast<f32x4*>(m + o) = mv;
                *reinterpret_cast<f32x4*>(v + o) = vv;
            }
            if (pb) *reinterpret_cast<bf16x4*>(pb + o) = bf16x4{(bf16)pv[0], (bf16)pv[1], (bf16)pv[2], (bf16)pv[3]};
            if (tp) {
                const f32x4 tv = *reinterpret_cast<f32x4*>(tp + o);
                const f32x4 tn = tv * ema_m + pv * (1.f - ema_m);
                *reinterpret_cast<f32x4*>(tp + o) = tn;
                if (tb) *reinterpret_cast<bf16x4*>(tb + o) = bf16x4{(bf16)tn[0], (bf16)tn[1], (bf16)tn[2], (bf16)tn[3]};
            }
        } else {
            for (long j = o; j < n; ++j) {
                float pe = p[j];
                if (has_grad) {
                    float me = m[j], ve = v[j];
                    adamw_elem(pe, g[j] * gscale, me, ve, decay, b1, b2, step_size, inv_sqrt_bc2, eps);
                    p[j] = pe;
                    m[j] = me;
                    v[j] = ve;
                }
                if (pb) pb[j] = (bf16)pe;
                if (tp) {
                    const float tn = tp[j] * ema_m + pe * (1.f - ema_m);
                    tp[j] = tn;
                    if (tb) tb[j] = (bf16)tn;
                }
            }
        }
    }
}

// SGD with momentum / LARS + EMA.  mu = momentum * mu + dp, p -= lr * mu with
//   SGD  (torch.optim.SGD, dampening 0, no nesterov): dp = c g + wd p              (wd: group 0 only, as the schedule sets it)
//   LARS (utils.py:533-557):                          dp = q (c g + wd p), q = eta |p| / |c g + wd p| for ndim != 1 (group 0)
// c = the per-tensor clip factor.  `stats` holds (sum g^2, sum p^2, sum g p) per tensor for LARS, sum g^2 for SGD.
template <bool LARS>
__global__ __launch_bounds__(256) void clip_momentum_ema_kernel(const long* __restrict__ tensors, const int* __restrict__ chunks,
                                                                const float* __restrict__ stats, int nstats, float clip, float lr, float wd,
                                                                float momentum, float eta, float ema_m, int* __restrict__ skipped) {
    if (stats_not_finite(stats, nstats, skipped)) return;
    const int tid = chunks[2 * blockIdx.x], ci = chunks[2 * blockIdx.x + 1];
    const long* tt = tensors + (long)tid * TFIELDS;
    float* p = reinterpret_cast<float*>(tt[0]);
    const float* g = reinterpret_cast<const float*>(tt[1]);
    float* mu = reinterpret_cast<float*>(tt[2]);
    float* tp = reinterpret_cast<float*>(tt[4]);
    bf16* pb = reinterpret_cast<bf16*>(tt[10]);
    bf16* tb = reinterpret_cast<bf16*>(tt[11]);
    const long n = tt[5];
    const bool has_grad = tt[7] & 1;
    const float wdp = (tt[6] == 0) ? wd : 0.f;
    const float* st = stats + (long)tid * (LARS ? 3 : 1);
    float c = 1.f, q = 1.f;
    if (has_grad) {
        if (clip > 0.f) {
            const float coef = clip / (sqrtf(st[0]) + 1e-6f);
            if (coef < 1.f) c = coef;
        }
        if (LARS && tt[6] == 0) {
            const float pn = sqrtf(st[1]);
            const float un = sqrtf(fmaxf(c * c * st[0] + 2.f * c * wdp * st[2] + wdp * wdp * st[1], 0.f));
            if (pn > 0.f && un > 0.f) q = eta * pn / un;
        }
    }
    const long base = (long)ci * CHUNK;
#pragma unroll
    for (int i = 0; i < CHUNK / 256 / 4; ++i) {
        const long o = base + (i * 256 + threadIdx.x) * 4;
        const int cnt = o + 4 <= n ? 4 : (o < n ? (int)(n - o) : 0);
        for (int e = 0; e < cnt; ++e) {  // (the update is a small fraction of the step: scalar tail-safe form)
            const long j = o + e;
            float pe = p[j];
            if (has_grad) {
                const float dp = q * (c * g[j] + wdp * pe);
                const float me = momentum * mu[j] + dp;
                mu[j] = me;
                pe -= lr * me;
                p[j] = pe;
            }
            if (pb) pb[j] = (bf16)pe;
            if (tp) {
                const float tn = tp[j] * ema_m + pe * (1.f - ema_m);
                tp[j] = tn;
                if (tb) tb[j] = (bf16)tn;
            }
        }
    }
}

}  // namespace

int esvit_i_update_chunk_elems() { return CHUNK; }  // esvit_query

extern "C" int esvit_grad_sqnorm(const int64_t* tensors, int ntensors, const int32_t* chunks, int nchunks, int stats, float* sqnorms,
                                 esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(tensors && chunks && sqnorms && ntensors > 0 && nchunks > 0 && (stats == 1 || stats == 3), "esvit_grad_sqnorm: bad args");
    hipError_t e = hipMemsetAsync(sqnorms, 0, (size_t)ntensors * stats * sizeof(float), stream);
    if (e != hipSuccess) {
        esvit_set_error("esvit_grad_sqnorm: memset failed: %s", hipGetErrorString(e));
        return ESVIT_ERR_HIP;
    }
    if (stats == 3) hipLaunchKernelGGL(grad_sqnorm_kernel<3>, dim3(nchunks), dim3(256), 0, stream, (const long*)tensors, chunks, sqnorms);
    else hipLaunchKernelGGL(grad_sqnorm_kernel<1>, dim3(nchunks), dim3(256), 0, stream, (const long*)tensors, chunks, sqnorms);
    ESVIT_CHECK_LAUNCH("grad_sqnorm");
    return ESVIT_OK;
}

extern "C" int esvit_fused_clip_update_ema(int rule, const int64_t* tensors, int ntensors, const int32_t* chunks, int nchunks,
                                           const float* sqnorms, float clip, float lr, float wd, float beta1, float beta2, float eps,
                                           float ema_m, int32_t* skipped, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(tensors && chunks && sqnorms && ntensors > 0 && nchunks > 0, "esvit_fused_clip_update_ema: bad args");
    ESVIT_CHECK_ARG(rule == ESVIT_RULE_ADAMW || rule == ESVIT_RULE_SGD || rule == ESVIT_RULE_LARS, "esvit_fused_clip_update_ema: bad rule %d", rule);
    if (rule == ESVIT_RULE_ADAMW)
        hipLaunchKernelGGL(clip_adamw_ema_kernel, dim3(nchunks), dim3(256), 0, stream, (const long*)tensors, chunks, sqnorms, ntensors, clip, lr, wd,
                           beta1, beta2, eps, ema_m, skipped);
    else if (rule == ESVIT_RULE_SGD)
        hipLaunchKernelGGL(clip_momentum_ema_kernel<false>, dim3(nchunks), dim3(256), 0, stream, (const long*)tensors, chunks, sqnorms, ntensors,
                           clip, lr, wd, beta1, beta2, ema_m, skipped);
    else
        hipLaunchKernelGGL(clip_momentum_ema_kernel<true>, dim3(nchunks), dim3(256), 0, stream, (const long*)tensors, chunks, sqnorms, 3 * ntensors,
                           clip, lr, wd, beta1, beta2, ema_m, skipped);
    ESVIT_CHECK_LAUNCH("fused_clip_update_ema");
    return ESVIT_OK;
}
