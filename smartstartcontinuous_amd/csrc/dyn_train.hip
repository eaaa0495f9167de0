// dyn_train.hip -- one Adam step of the NND_MB dynamics model on a mini-batch
// (NN_Dynamics_Model/dynamics_model.py:41-50, 98-113) for ANY feedforward_network shape.
//
// fp32 throughout (the reference trains in fp64 on the CPU; parity tolerance in the tests).  The step is
// a chain of small launches over the batch (512 rows in the shipped runs): gather -> forward layers
// (reusing the fp32 layer kernel of dyn_model.hip) -> output delta -> per layer {dW, db, dX, ReLU mask,
// Adam}.  The bias-corrected step size is computed on the device from a device-side step counter, so
// consecutive steps can be enqueued without a host round trip.
#include "ssc_device.h"
#include "ssc_host.h"

namespace ssc {

// dyn_model.hip
void launch_mlp_layer_f32(bool relu, int64_t m, int K, int N, const float *X, const float *W, const float *b, float *Y,
                          hipStream_t s);

__global__ __launch_bounds__(256) void train_gather_kernel(int B, int in, int out, const float *__restrict__ X,
                                                           const float *__restrict__ Z, const int32_t *__restrict__ idx,
                                                           float *__restrict__ xb, float *__restrict__ zb) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < B * in) {
        const int r = e / in, c = e - r * in;
        xb[e] = X[(int64_t)idx[r] * in + c];
    }
    if (e < B * out) {
        const int r = e / out, c = e - r * out;
        zb[e] = Z[(int64_t)idx[r] * out + c];
    }
}

// scal[0] = lr_t for this step, scal[1] = loss accumulator (zeroed); t += 1
__global__ void train_begin_kernel(int32_t *t, float lr, float b1, float b2, float *scal) {
    const int tt = t[0] + 1;
    t[0] = tt;
    scal[0] = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)tt)) / (1.0 - pow((double)b1, (double)tt)));
    scal[1] = 0.0f;
}

// dY = 2 (y - z) / (B * out)   (d mean((z - y)^2) / dy, dynamics_model.py:41); loss accumulated
__global__ __launch_bounds__(256) void train_out_delta_kernel(int n, const float *__restrict__ y,
                                                              const float *__restrict__ z, float *__restrict__ dy,
                                                              float *__restrict__ scal) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    float l = 0.0f;
    if (e < n) {
        const float d = y[e] - z[e];
        dy[e] = 2.0f * d / (float)n;
        l = d * d / (float)n;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) l += __shfl_xor(l, m);
    if ((threadIdx.x & 63) == 0) atomicAdd(scal + 1, l);
}

// one thread per weight: g = sum_b A[b][i] dZ[b][j], then Adam in place
__global__ __launch_bounds__(256) void train_weight_kernel(int B, int K, int N, const float *__restrict__ A,
                                                           const float *__restrict__ dZ, float *__restrict__ W,
                                                           float *__restrict__ mW, float *__restrict__ vW,
                                                           float *__restrict__ bias, float *__restrict__ mb,
                                                           float *__restrict__ vb, const float *__restrict__ scal,
                                                           float b1, float b2, float eps) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const float lr_t = scal[0];
    if (e < K * N) {
        const int i = e / N, j = e - i * N;
        float g = 0.0f;
        for (int b = 0; b < B; ++b) g = fmaf(A[(int64_t)b * K + i], dZ[(int64_t)b * N + j], g);
        const float m = b1 * mW[e] + (1.0f - b1) * g;
        const float v = b2 * vW[e] + (1.0f - b2) * g * g;
        mW[e] = m; vW[e] = v;
        W[e] -= lr_t * m / (sqrtf(v) + eps);
    } else if (e < K * N + N) {
        const int j = e - K * N;
        float g = 0.0f;
        for (int b = 0; b < B; ++b) g += dZ[(int64_t)b * N + j];
        const float m = b1 * mb[j] + (1.0f - b1) * g;
        const float v = b2 * vb[j] + (1.0f - b2) * g * g;
        mb[j] = m; vb[j] = v;
        bias[j] -= lr_t * m / (sqrtf(v) + eps);
    }
}

// dA[b][i] = (A[b][i] > 0) * sum_j dZ[b][j] W[i][j]    (ReLU of feedforward_network.py:19)
__global__ __launch_bounds__(256) void train_back_kernel(int B, int K, int N, const float *__restrict__ dZ,
                                                         const float *__restrict__ W, const float *__restrict__ A,
                                                         float *__restrict__ dA) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * K) return;
    const int b = e / K, i = e - b * K;
    float g = 0.0f;
    for (int j = 0; j < N; ++j) g = fmaf(dZ[(int64_t)b * N + j], W[(int64_t)i * N + j], g);
    dA[e] = (A[e] > 0.0f) ? g : 0.0f;
}

__global__ void train_loss_out_kernel(const float *scal, float *loss) { loss[0] = scal[1]; }

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace ssc

using namespace ssc;

extern "C" {

size_t ssc_mlp_train_workspace_bytes(const ssc_mlp_train_desc *net, int32_t B) {
    if (net == nullptr || B <= 0 || net->n_layers < 1 || net->n_layers > SSC_MAX_LAYERS) return 0;
    size_t total = 256;  // scalars
    for (int l = 0; l <= net->n_layers; ++l) total += al256((size_t)B * net->dims[l] * 4);  // activations (0 = x batch)
    int maxw = 0;
    for (int l = 0; l <= net->n_layers; ++l) maxw = net->dims[l] > maxw ? net->dims[l] : maxw;
    total += al256((size_t)B * net->dims[net->n_layers] * 4);  // z batch
    total += 2 * al256((size_t)B * maxw * 4);                  // delta ping-pong
    return total;
}

int ssc_mlp_train_step(const ssc_mlp_train_desc *net, const float *d_X, const float *d_Z, const int32_t *d_idx,
                       int32_t B, float *d_loss, void *d_workspace, size_t workspace_bytes, ssc_stream_t stream) {
    SSC_REQUIRE(net != nullptr, "ssc_mlp_train_step: net NULL");
    SSC_REQUIRE(net->n_layers >= 1 && net->n_layers <= SSC_MAX_LAYERS, "ssc_mlp_train_step: bad n_layers");
    SSC_REQUIRE(B >= 1 && B <= 65536, "ssc_mlp_train_step: batch %d out of range", B);
    for (int l = 0; l <= net->n_layers; ++l)
        SSC_REQUIRE(net->dims[l] >= 1 && net->dims[l] <= 8192, "ssc_mlp_train_step: bad dims[%d]", l);
    for (int l = 0; l < net->n_layers; ++l)
        SSC_REQUIRE(net->W[l] && net->b[l] && net->mW[l] && net->vW[l] && net->mb[l] && net->vb[l],
                    "ssc_mlp_train_step: NULL parameter / moment pointer (layer %d)", l);
    SSC_REQUIRE(net->adam_t && d_X && d_Z && d_idx, "ssc_mlp_train_step: NULL pointer");
    const size_t need = ssc_mlp_train_workspace_bytes(net, B);
    SSC_REQUIRE(d_workspace && workspace_bytes >= need, "ssc_mlp_train_step: workspace %zu < %zu", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    const int L = net->n_layers;
    char *w = static_cast<char *>(d_workspace);
    float *scal = reinterpret_cast<float *>(w); w += 256;
    float *act[SSC_MAX_LAYERS + 1];
    for (int l = 0; l <= L; ++l) { act[l] = reinterpret_cast<float *>(w); w += al256((size_t)B * net->dims[l] * 4); }
    float *zb = reinterpret_cast<float *>(w); w += al256((size_t)B * net->dims[L] * 4);
    int maxw = 0;
    for (int l = 0; l <= L; ++l) maxw = net->dims[l] > maxw ? net->dims[l] : maxw;
    float *d0 = reinterpret_cast<float *>(w); w += al256((size_t)B * maxw * 4);
    float *d1 = reinterpret_cast<float *>(w);
    const int in = net->dims[0], out = net->dims[L];
    hipLaunchKernelGGL(train_begin_kernel, dim3(1), dim3(1), 0, s, net->adam_t, net->lr, net->beta1, net->beta2, scal);
    hipLaunchKernelGGL(train_gather_kernel, dim3(blocks_for((int64_t)B * (in > out ? in : out))), dim3(256), 0, s, B, in,
                       out, d_X, d_Z, d_idx, act[0], zb);
    for (int l = 0; l < L; ++l)
        launch_mlp_layer_f32(l != L - 1, B, net->dims[l], net->dims[l + 1], act[l], net->W[l], net->b[l], act[l + 1], s);
    hipLaunchKernelGGL(train_out_delta_kernel, dim3(blocks_for((int64_t)B * out)), dim3(256), 0, s, B * out, act[L], zb,
                       d0, scal);
    float *dz = d0, *dprev = d1;
    for (int l = L - 1; l >= 0; --l) {
        const int K = net->dims[l], N = net->dims[l + 1];
        if (l > 0)  // delta of the previous layer from the OLD weights, before Adam touches them
            hipLaunchKernelGGL(train_back_kernel, dim3(blocks_for((int64_t)B * K)), dim3(256), 0, s, B, K, N, dz,
                               net->W[l], act[l], dprev);
        hipLaunchKernelGGL(train_weight_kernel, dim3(blocks_for((int64_t)K * N + N)), dim3(256), 0, s, B, K, N, act[l],
                           dz, net->W[l], net->mW[l], net->vW[l], net->b[l], net->mb[l], net->vb[l], scal, net->beta1,
                           net->beta2, net->epsilon);
        float *t = dz; dz = dprev; dprev = t;
    }
    if (d_loss != nullptr) hipLaunchKernelGGL(train_loss_out_kernel, dim3(1), dim3(1), 0, s, scal, d_loss);
    return check_launch("ssc_mlp_train_step");
}

}  // extern "C"
