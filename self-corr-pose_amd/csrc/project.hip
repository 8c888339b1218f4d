// self-corr-pose_amd/csrc/project.hip -- camera projection of the predicted vertices, forward and backward, one launch each.
//
// Replaces model/util/loss_utils.py:38-61 as the trainer calls it for every render pass (render(): verts.bmm(rot) + trans, pinhole_cam,
// y flipped) and for the projected vertex positions of model/module/renderer.py:63-67:
//     cam = verts @ R + t            (row vectors; [B,V,3] x [B,3,3] + [B,1,3])
//     x = pp_x + cam_x * f_x / cam_z,   y = pp_y + cam_y * f_y / cam_z   (evaluated in float64 and rounded when the intrinsics are
//                                        float64 -- the data loader's are: the reference assigns the promoted expression into a float32
//                                        tensor, loss_utils.py:40-41 -- else in float32)
//     out = (x, -y, cam_z)            (flip_y = 0: (x, y, cam_z), the projected vertex positions)
// As torch ops this is ~15 launches forward and, because every `verts[:, :, k]` select has a zero-fill + copy backward, ~35 backward --
// three times per step, all on the step's serial chain.  The K = 3 product is evaluated as ((v0 r0j + v1 r1j) + v2 r2j) + t_j without
// contraction: a FIXED order (a library GEMM picks its own per solution, which made the sigma = 1e-4 silhouette depend on which
// solution a process had tuned).  Backward: partials in float64, rounded once; the rotation / translation gradients are sums over the
// vertices of an image, reduced by one workgroup per image in a fixed order (deterministic).
// Built with -ffp-contract=off (build.py).  HBM-bound, 12 B in + 12 B out per vertex: nothing to tune.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {
template <bool F64>
__global__ __launch_bounds__(256) void project_forward_kernel(const float* __restrict__ verts, const float* __restrict__ rot,
                                                              const float* __restrict__ trans, const void* __restrict__ foc,
                                                              const void* __restrict__ pp, int V, int flip_y, float* __restrict__ out,
                                                              float* __restrict__ cam_out) {
    const int b = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float* r = rot + (size_t)b * 9;
    const float* p = verts + ((size_t)b * V + v) * 3;
    const float v0 = p[0], v1 = p[1], v2 = p[2];
    float cam[3];
#pragma unroll
    for (int j = 0; j < 3; j++) cam[j] = ((v0 * r[j] + v1 * r[3 + j]) + v2 * r[6 + j]) + trans[b * 3 + j];
    float x, y;
    if (F64) {
        const double* f = static_cast<const double*>(foc) + b * 2;
        const double* c = static_cast<const double*>(pp) + b * 2;
        x = (float)(c[0] + (double)cam[0] * f[0] / (double)cam[2]);
        y = (float)(c[1] + (double)cam[1] * f[1] / (double)cam[2]);
    } else {
        const float* f = static_cast<const float*>(foc) + b * 2;
        const float* c = static_cast<const float*>(pp) + b * 2;
        x = c[0] + cam[0] * f[0] / cam[2];
        y = c[1] + cam[1] * f[1] / cam[2];
    }
    float* o = out + ((size_t)b * V + v) * 3;
    o[0] = x;
    o[1] = flip_y ? -y : y;
    o[2] = cam[2];
    if (cam_out) {
        float* c = cam_out + ((size_t)b * V + v) * 3;
        c[0] = cam[0]; c[1] = cam[1]; c[2] = cam[2];
    }
}

// one workgroup per image: g_verts per vertex, g_rot / g_trans reduced over the image's vertices
template <bool F64>
__global__ __launch_bounds__(256) void project_backward_kernel(const float* __restrict__ g_out, const float* __restrict__ verts,
                                                               const float* __restrict__ rot, const float* __restrict__ cam,
                                                               const void* __restrict__ foc, int V, int flip_y,
                                                               float* __restrict__ g_verts, float* __restrict__ g_rot,
                                                               float* __restrict__ g_trans) {
    const int b = blockIdx.x, tid = threadIdx.x;
    double fx, fy;
    if (F64) { fx = static_cast<const double*>(foc)[b * 2]; fy = static_cast<const double*>(foc)[b * 2 + 1]; }
    else { fx = static_cast<const float*>(foc)[b * 2]; fy = static_cast<const float*>(foc)[b * 2 + 1]; }
    const float* r = rot + (size_t)b * 9;
    double acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = 0.0;
    for (int v = tid; v < V; v += 256) {
        const size_t o = ((size_t)b * V + v) * 3;
        const double gx = g_out[o], gy = flip_y ? -(double)g_out[o + 1] : (double)g_out[o + 1], gz = g_out[o + 2];
        const double c0 = cam[o], c1 = cam[o + 1], c2 = cam[o + 2];
        const double iz = 1.0 / c2;
        double d[3];
        d[0] = gx * fx * iz;
        d[1] = gy * fy * iz;
        d[2] = gz - (gx * c0 * fx + gy * c1 * fy) * iz * iz;
        const double p[3] = {verts[o], verts[o + 1], verts[o + 2]};
        if (g_verts) {
#pragma unroll
            for (int i = 0; i < 3; i++) g_verts[o + i] = (float)((d[0] * r[3 * i] + d[1] * r[3 * i + 1]) + d[2] * r[3 * i + 2]);
        }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) acc[3 * i + j] += p[i] * d[j];
#pragma unroll
        for (int j = 0; j < 3; j++) acc[9 + j] += d[j];
    }
    // fixed-order tree over the 256 threads (LDS), thread 0 writes
    __shared__ double red[256];
    for (int i = 0; i < 12; i++) {
        red[tid] = acc[i];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) {
            if (i < 9) { if (g_rot) g_rot[b * 9 + i] = (float)red[0]; }
            else if (g_trans) g_trans[b * 3 + (i - 9)] = (float)red[0];
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int scp_project_vertices_forward(const float* verts, const float* rot, const float* trans, const void* foc, const void* pp,
                                            int intrinsics_f64, int flip_y, int B, int V, float* out, float* cam, void* stream) {
    if (B <= 0 || V <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((V + 255) / 256, B);
    if (intrinsics_f64) hipLaunchKernelGGL(project_forward_kernel<true>, grid, dim3(256), 0, st, verts, rot, trans, foc, pp, V, flip_y, out, cam);
    else hipLaunchKernelGGL(project_forward_kernel<false>, grid, dim3(256), 0, st, verts, rot, trans, foc, pp, V, flip_y, out, cam);
    return scp::check_launch("project_vertices_forward");
}

extern "C" int scp_project_vertices_backward(const float* g_out, const float* verts, const float* rot, const float* cam, const void* foc,
                                             int intrinsics_f64, int flip_y, int B, int V, float* g_verts, float* g_rot, float* g_trans,
                                             void* stream) {
    if (B <= 0 || V <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (intrinsics_f64) hipLaunchKernelGGL(project_backward_kernel<true>, dim3(B), dim3(256), 0, st, g_out, verts, rot, cam, foc, V, flip_y, g_verts, g_rot, g_trans);
    else hipLaunchKernelGGL(project_backward_kernel<false>, dim3(B), dim3(256), 0, st, g_out, verts, rot, cam, foc, V, flip_y, g_verts, g_rot, g_trans);
    return scp::check_launch("project_vertices_backward");
}
