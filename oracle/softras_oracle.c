/*
 * oracle/softras_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C, fp32, no FMA contraction) of the SoftRas soft rasteriser that the
 * reference trains with.  It exists only so that tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py can check / time the HIP kernels against it.  Nothing under
 * self-corr-pose_amd/ may link, import or call it.
 *
 * Algorithm restated from (all paths relative to /root/reference):
 *   third-party/softras/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu
 *     :245-305  per-face precompute (inverse of [x y 1], Gram matrix + 1, obtuse-corner flag)
 *     :24-151   barycentric weights, bbox reject, euclidean point-to-triangle distance
 *     :308-483  per-pixel forward (sigmoid coverage, alpha prod/sum/hard, z-buffer or online softmax)
 *     :486-668  per-pixel backward (d alpha, d softmax-rgb, d z, distance Jacobian, texture grads)
 * The arithmetic follows the reference expression by expression, including the places where a
 * double literal promotes an fp32 sub-expression to fp64 before it is narrowed again
 * (SURVEY.md F12 / Appendix A), because the gamma=1e-4 depth softmax amplifies 1-ulp changes.
 *
 * Pinning: tests/test_oracle_golden.py checks this file bit-for-bit (forward) and to fp32
 * round-off (backward sums) against tests/golden/softras_*.npz, which were produced in the build
 * container by tests/golden/make_golden.py from the reference kernel bodies themselves.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stddef.h>

typedef struct {
    int batch, nfaces, size, tex_size, tex_res;
    float near_, far_, eps, sigma, dist_eps, gamma;
    int dist_mode;   /* 0 hard, 1 barycentric, 2 euclidean   (soft_rasterize.py:22) */
    int rgb_mode;    /* 0 hard z-buffer, 1 softmax            (soft_rasterize.py:23) */
    int alpha_mode;  /* 0 hard, 1 sum, 2 prod                 (soft_rasterize.py:24) */
    int sample_mode; /* 0 surface, 1 vertex                   (soft_rasterize.py:25) */
    int double_side;
} sr_params;

static inline float f_max3(float a, float b, float c) { float m = a > b ? a : b; return m > c ? m : c; }
static inline float f_min3(float a, float b, float c) { float m = a < b ? a : b; return m < c ? m : c; }
/* CUDA's mixed overloads max(float,double)/min(float,double) evaluate in double */
static inline double d_max(double a, double b) { return a > b ? a : b; }
static inline double d_min(double a, double b) { return a < b ? a : b; }

/* ---- kernel.cu:245-305 ------------------------------------------------------------------- */
static void face_precompute(const float *v, float *info)
{
    const float x0 = v[0], y0 = v[1], x1 = v[3], y1 = v[4], x2 = v[6], y2 = v[7];
    const float adj[9] = {
        y1 - y2, x2 - x1, x1 * y2 - x2 * y1,
        y2 - y0, x0 - x2, x2 * y0 - x0 * y2,
        y0 - y1, x1 - x0, x0 * y1 - x1 * y0 };
    float det = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0);
    det = det > 0 ? (float)d_max(det, 1e-10) : (float)d_min(det, -1e-10);
    for (int k = 0; k < 9; k++) info[k] = adj[k] / det;
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++)
            info[9 + 3 * j + k] = v[3 * j] * v[3 * k] + v[3 * j + 1] * v[3 * k + 1] + 1;
    const float px[3] = { x0, x1, x2 }, py[3] = { y0, y1, y2 };
    for (int k = 0; k < 3; k++) {
        const int a = (k + 1) % 3, b = (k + 2) % 3;
        if ((px[a] - px[k]) * (px[b] - px[k]) + (py[a] - py[k]) * (py[b] - py[k]) < 0) {
            info[18 + k] = 1;
            break;
        }
    }
}

/* ---- kernel.cu:32-38 --------------------------------------------------------------------- */
static inline int outside_bbox(float x, float y, const float *v, float margin)
{
    return x > f_max3(v[0], v[3], v[6]) + margin || x < f_min3(v[0], v[3], v[6]) - margin ||
           y > f_max3(v[1], v[4], v[7]) + margin || y < f_min3(v[1], v[4], v[7]) - margin;
}

/* ---- kernel.cu:47-50 --------------------------------------------------------------------- */
static inline int weights_inside(const float *w)
{
    return w[0] <= 1 && w[0] >= 0 && w[1] <= 1 && w[1] >= 0 && w[2] <= 1 && w[2] >= 0;
}

/* ---- kernel.cu:41-44 --------------------------------------------------------------------- */
static inline int front_facing(const float *v)
{
    return (v[7] - v[1]) * (v[3] - v[0]) < (v[4] - v[1]) * (v[6] - v[0]);
}

/* ---- kernel.cu:53-58 --------------------------------------------------------------------- */
static inline void clip_weights(float *w)
{
    for (int k = 0; k < 3; k++) w[k] = (float)d_max(d_min(w[k], 1.), 0.);
    const float s = (float)d_max(w[0] + w[1] + w[2], 1e-5);
    for (int k = 0; k < 3; k++) w[k] /= s;
}

/* one edge projection shared by the inside and outside branches (kernel.cu:81-87 / :132-138) */
static inline float edge_param(const float *sym, const float *w, int v0, int v1)
{
    const float a0 = sym[3 * v0 + 0] - sym[3 * v1 + 0];
    const float a1 = sym[3 * v0 + 1] - sym[3 * v1 + 1];
    const float a2 = sym[3 * v0 + 2] - sym[3 * v1 + 2];
    const float a[3] = { a0, a1, a2 };
    return (w[0] * a0 + w[1] * a1 + w[2] * a2 - a[v1]) / (a[v0] - a[v1]);
}

/* ---- kernel.cu:61-151 -------------------------------------------------------------------- */
static void euclid_distance(float *sign, float *dx, float *dy, const float *w, float *t,
                            const float *v, const float *info, float xp, float yp)
{
    const float *sym = info + 9, *obt = info + 18;
    if (w[0] > 0 && w[1] > 0 && w[2] > 0 && w[0] < 1 && w[1] < 1 && w[2] < 1) {
        float best = 100000000, bx = 0, by = 0;
        for (int k = 0; k < 3; k++) {
            const int v0 = k, v1 = (k + 1) % 3, v2 = (k + 2) % 3;
            float t0[3];
            t0[v0] = edge_param(sym, w, v0, v1);
            t0[v1] = 1 - t0[v0];
            t0[v2] = 0;
            t0[0] -= w[0]; t0[1] -= w[1]; t0[2] -= w[2];
            const float ex = t0[0] * v[0] + t0[1] * v[3] + t0[2] * v[6];
            const float ey = t0[0] * v[1] + t0[1] * v[4] + t0[2] * v[7];
            const float d = ex * ex + ey * ey;
            if (d < best) { best = d; bx = ex; by = ey; t[0] = t0[0]; t[1] = t0[1]; t[2] = t0[2]; }
        }
        *dx = bx; *dy = by; *sign = 1;
    } else {
        int v0 = -1;
        if (w[1] <= 0 && w[2] <= 0) {
            v0 = 0;
            if (obt[0] == 1 && (xp - v[0]) * (v[6] - v[0]) + (yp - v[1]) * (v[7] - v[1]) > 0) v0 = 2;
        } else if (w[2] <= 0 && w[0] <= 0) {
            v0 = 1;
            if (obt[1] == 1 && (xp - v[3]) * (v[0] - v[3]) + (yp - v[4]) * (v[1] - v[4]) > 0) v0 = 0;
        } else if (w[0] <= 0 && w[1] <= 0) {
            v0 = 2;
            if (obt[2] == 1 && (xp - v[6]) * (v[3] - v[6]) + (yp - v[7]) * (v[4] - v[7]) > 0) v0 = 1;
        } else if (w[0] <= 0) v0 = 1;
        else if (w[1] <= 0) v0 = 2;
        else if (w[2] <= 0) v0 = 0;
        /* v0 == -1 (all w>0, one >=1 by rounding) indexes before face_sym in the reference
           (SURVEY App. A.2c); here it is mapped to edge 0 -- fixtures never reach it. */
        if (v0 < 0) v0 = 0;
        const int v1 = (v0 + 1) % 3, v2 = (v0 + 2) % 3;
        t[v0] = edge_param(sym, w, v0, v1);
        t[v1] = 1 - t[v0];
        t[v2] = 0;
        for (int k = 0; k < 3; k++) {
            t[k] = (float)d_min(d_max(t[k], 0.), 1.);
            t[k] -= w[k];
        }
        *dx = t[0] * v[0] + t[1] * v[3] + t[2] * v[6];
        *dy = t[0] * v[1] + t[1] * v[4] + t[2] * v[7];
        *sign = -1;
    }
}

/* ---- kernel.cu:154-158 ------------------------------------------------------------------- */
static inline float bary_distance(const float *w)
{
    float d = w[0] > w[1] ? (w[1] > w[2] ? w[2] : w[1]) : (w[0] > w[2] ? w[2] : w[0]);
    return d > 0 ? d * d : -(d * d);
}

/* ---- kernel.cu:178-194 ------------------------------------------------------------------- */
static inline float sample_texture(const float *tex, const float *w, int R, int k, int mode)
{
    if (mode == 0) {
        /* when a clipped weight is exactly 1 this indexes past the face's own R*R texels, into the
           next face's texture -- reproduced as is; the caller guarantees the batch's last face is
           never sampled that way (the reference reads out of bounds there) */
        const int wx = (int)(w[0] * R), wy = (int)(w[1] * R);
        if ((w[0] + w[1]) * R - wx - wy <= 1) return tex[(wy * R + wx) * 3 + k];
        return tex[((R - 1 - wy) * R + (R - 1 - wx)) * 3 + k];
    }
    return w[0] * tex[k] + w[1] * tex[3 + k] + w[2] * tex[6 + k];
}

/* ---- kernel.cu:197-217 ------------------------------------------------------------------- */
static inline float sample_texture_grad(float g, const float *w, int R, int j, int mode)
{
    if (mode == 0) {
        const int wx = (int)(w[0] * R), wy = (int)(w[1] * R);
        if ((w[0] + w[1]) * R - wx - wy <= 1) return j == wy * R + wx ? g : 0.f;
        return j == (R - 1 - wy) * R + (R - 1 - wx) ? g : 0.f;
    }
    return w[j] * g;
}

/* The part of the per-(pixel,face) work that forward and backward share (kernel.cu:375-404 ==
 * :543-572).  Returns 0 when the pair is skipped. */
typedef struct { float w[3], t[3], sign, dx, dy, dis, frag; } coverage;

static int pair_coverage(const sr_params *p, float xp, float yp, float threshold, float margin,
                         const float *v, const float *info, coverage *c)
{
    if (outside_bbox(xp, yp, v, margin)) return 0;
    for (int k = 0; k < 3; k++) c->w[k] = info[3 * k] * xp + info[3 * k + 1] * yp + info[3 * k + 2];
    c->sign = 0; c->dx = 0; c->dy = 0; c->dis = 0; c->t[0] = c->t[1] = c->t[2] = 0;
    if (p->dist_mode == 0) {
        c->frag = weights_inside(c->w) ? 1.f : 0.f;
        if (c->frag == 0.f) return 0;
    } else if (p->dist_mode == 1) {
        c->dis = bary_distance(c->w);
        for (int k = 0; k < 3; k++) c->t[k] = c->w[k];
        if (-c->dis >= threshold) return 0;
        c->frag = (float)(1. / (1. + expf(-c->dis / p->sigma)));
    } else {
        euclid_distance(&c->sign, &c->dx, &c->dy, c->w, c->t, v, info, xp, yp);
        c->dis = c->dx * c->dx + c->dy * c->dy;
        if (c->sign < 0 && c->dis >= threshold) return 0;
        c->frag = (float)(1. / (1. + expf(-c->sign * c->dis / p->sigma)));
    }
    return 1;
}

/* ---- kernel.cu:674-746 (launcher) + :245-305 + :308-483 ------------------------------------ */
void sr_oracle_forward(const float *faces, const float *textures, float *faces_info,
                       float *aggrs_info, float *soft_colors, const sr_params *p)
{
    const int B = p->batch, F = p->nfaces, S = p->size, T = p->tex_size, R = p->tex_res;
    const long npix = (long)S * S;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)B * F; i++) face_precompute(faces + 9 * i, faces_info + 27 * i);

    const float threshold = p->dist_eps * p->sigma;
    const float margin = sqrtf(threshold);
#pragma omp parallel for schedule(dynamic, 256)
    for (long i = 0; i < (long)B * npix; i++) {
        const int bn = (int)(i / npix);
        const long pn = i % npix;
        const int yi = S - 1 - (int)(pn / S), xi = (int)(pn % S);
        const float yp = (float)((2. * yi + 1. - S) / S);
        const float xp = (float)((2. * xi + 1. - S) / S);
        float *out = soft_colors + (long)bn * 4 * npix + pn;

        float col[4] = { 1.f, 1.f, 1.f, 0.f };
        if (p->alpha_mode == 2) col[3] = 1.f;
        float sm_sum = expf(p->eps / p->gamma);
        float sm_max = p->eps;
        for (int k = 0; k < 3; k++) {
            if (p->rgb_mode == 0) col[k] = out[k * npix];
            else if (p->rgb_mode == 1) col[k] = out[k * npix] * sm_sum;
        }
        float zmin = 10000000;
        int fmin = -1;

        for (int fn = 0; fn < F; fn++) {
            const float *v = faces + ((long)bn * F + fn) * 9;
            const float *info = faces_info + ((long)bn * F + fn) * 27;
            const float *tex = textures + ((long)bn * F + fn) * T * 3;
            coverage c;
            if (!pair_coverage(p, xp, yp, threshold, margin, v, info, &c)) continue;

            if (p->alpha_mode == 0) { if (c.frag > 0.5) col[3] = 1.f; }
            else if (p->alpha_mode == 1) col[3] += c.frag;
            else if (p->alpha_mode == 2) col[3] = (float)(col[3] * (1. - c.frag));

            float wc[3] = { c.w[0], c.w[1], c.w[2] };
            clip_weights(wc);
            const float zp = (float)(1. / (wc[0] / v[2] + wc[1] / v[5] + wc[2] / v[8]));
            if (zp < p->near_ || zp > p->far_) continue;

            if (p->rgb_mode == 0) {
                if (zp < zmin && weights_inside(c.w) && (p->double_side || front_facing(v))) {
                    zmin = zp;
                    fmin = fn;
                    for (int k = 0; k < 3; k++) col[k] = sample_texture(tex, wc, R, k, p->sample_mode);
                }
            } else if (p->rgb_mode == 1) {
                if (front_facing(v) || p->double_side) {
                    const float zn = (p->far_ - zp) / (p->far_ - p->near_);
                    float rescale = 1.f;
                    if (zn > sm_max) {
                        rescale = expf((sm_max - zn) / p->gamma);
                        sm_max = zn;
                    }
                    const float ez = expf((zn - sm_max) / p->gamma);
                    sm_sum = rescale * sm_sum + ez * c.frag;
                    for (int k = 0; k < 3; k++) {
                        const float ck = sample_texture(tex, wc, R, k, p->sample_mode);
                        col[k] = rescale * col[k] + ez * c.frag * ck;
                    }
                }
            }
        }

        if (p->alpha_mode == 0) out[3 * npix] = col[3];
        else if (p->alpha_mode == 1) out[3 * npix] = col[3] / F;
        else if (p->alpha_mode == 2) out[3 * npix] = (float)(1. - col[3]);

        float *ag = aggrs_info + (long)bn * 2 * npix + pn;
        if (p->rgb_mode == 0) {
            if (fmin != -1) for (int k = 0; k < 3; k++) out[k * npix] = col[k];
            ag[0] = zmin;
            ag[npix] = (float)fmin;
        } else if (p->rgb_mode == 1) {
            for (int k = 0; k < 3; k++) out[k * npix] = col[k] / sm_sum;
            ag[0] = sm_sum;
            ag[npix] = sm_max;
        }
    }
}

/* ---- kernel.cu:749-813 (launcher) + :486-668 ------------------------------------------------
 * Images are processed in parallel, pixels of one image in raster order, so the accumulation order
 * equals a sequential replay of the reference's atomicAdd stream (deterministic). */
void sr_oracle_backward(const float *faces, const float *textures, const float *soft_colors,
                        const float *faces_info, const float *aggrs_info, float *grad_faces,
                        float *grad_textures, const float *grad_soft_colors, const sr_params *p)
{
    const int B = p->batch, F = p->nfaces, S = p->size, T = p->tex_size, R = p->tex_res;
    const long npix = (long)S * S;
    const float threshold = p->dist_eps * p->sigma;
    const float margin = sqrtf(threshold);
#pragma omp parallel for schedule(dynamic, 1)
    for (int bn = 0; bn < B; bn++) {
        for (long pn = 0; pn < npix; pn++) {
            const int yi = S - 1 - (int)(pn / S), xi = (int)(pn % S);
            const float yp = (float)((2. * yi + 1 - S) / S);
            const float xp = (float)((2. * xi + 1 - S) / S);
            const float *img = soft_colors + (long)bn * 4 * npix + pn;
            const float *gimg = grad_soft_colors + (long)bn * 4 * npix + pn;
            const float sm_sum = aggrs_info[((long)bn * 2 + 0) * npix + pn];
            const float sm_max = aggrs_info[((long)bn * 2 + 1) * npix + pn];

            for (int fn = 0; fn < F; fn++) {
                const float *v = faces + ((long)bn * F + fn) * 9;
                const float *info = faces_info + ((long)bn * F + fn) * 27;
                const float *tex = textures + ((long)bn * F + fn) * T * 3;
                float *gv_out = grad_faces + ((long)bn * F + fn) * 9;
                float *gt_out = grad_textures + ((long)bn * F + fn) * T * 3;
                coverage c;
                if (!pair_coverage(p, xp, yp, threshold, margin, v, info, &c)) continue;

                float gv[3][3] = { { 0 } };
                float c_xy = 0;
                float c_alpha = gimg[3 * npix];
                if (p->alpha_mode == 1) c_alpha /= F;
                else if (p->alpha_mode == 2)
                    c_alpha = (float)(c_alpha * ((1 - img[3 * npix]) / d_max(1 - c.frag, 1e-6)));
                c_xy += c_alpha;

                float w0[3] = { c.w[0], c.w[1], c.w[2] };
                float w[3] = { c.w[0], c.w[1], c.w[2] };
                clip_weights(w);
                const float zp = (float)(1. / (w[0] / v[2] + w[1] / v[5] + w[2] / v[8]));
                if (zp < p->near_ || zp > p->far_) continue;

                if (p->rgb_mode == 0) {
                    if (fn == sm_max) {
                        for (int k = 0; k < 3; k++)
                            for (int j = 0; j < T; j++)
                                gt_out[3 * j + k] += sample_texture_grad(gimg[k * npix], w, R, j, p->sample_mode);
                    }
                } else if (p->rgb_mode == 1 && (front_facing(v) || p->double_side)) {
                    float c_rgb = 0.f;
                    const float zn = (p->far_ - zp) / (p->far_ - p->near_);
                    const float zs = c.frag * expf((zn - sm_max) / p->gamma) / sm_sum;
                    for (int k = 0; k < 3; k++) {
                        const float gk = gimg[k * npix];
                        for (int j = 0; j < T; j++)
                            gt_out[3 * j + k] += zs * sample_texture_grad(gk, w, R, j, p->sample_mode);
                        const float ck = sample_texture(tex, w, R, k, p->sample_mode);
                        c_rgb += gk * (ck - img[k * npix]);
                    }
                    c_rgb *= zs;
                    c_xy += c_rgb / c.frag;
                    const float c_z = c_rgb / p->gamma / (p->near_ - p->far_) * zp * zp;
                    gv[0][2] = c_z * w[0] / v[2] / v[2];
                    gv[1][2] = c_z * w[1] / v[5] / v[5];
                    gv[2][2] = c_z * w[2] / v[8] / v[8];
                }

                c_xy *= c.frag * (1 - c.frag) / p->sigma;
                if (p->dist_mode == 1) {
                    /* kernel.cu:161-175 (t holds the unclipped weights in this mode) */
                    const float *t = c.t;
                    const int pm = t[0] > t[1] ? (t[1] > t[2] ? 2 : 1) : (t[0] > t[2] ? 2 : 0);
                    for (int l = 0; l < 2; l++)
                        for (int k = 0; k < 3; k++) {
                            float g = 0;
                            for (int q = 0; q < 3; q++)
                                g += -info[3 * pm + l] * info[3 * k + q] * (q == 0 ? xp : (q == 1 ? yp : 1));
                            gv[k][l] = g * c_xy;
                            gv[k][l] = (float)(gv[k][l] * (c.dis > 0 ? (2. * sqrtf(c.dis)) : (2. * sqrtf(-c.dis))));
                        }
                } else if (p->dist_mode == 2) {
                    for (int k = 0; k < 3; k++)
                        for (int l = 0; l < 2; l++)
                            gv[k][l] = 2 * c.sign * c_xy * (c.t[k] + w0[k]) * (l == 0 ? c.dx : c.dy);
                }

                gv_out[0] += gv[0][0]; gv_out[1] += gv[0][1];
                gv_out[3] += gv[1][0]; gv_out[4] += gv[1][1];
                gv_out[6] += gv[2][0]; gv_out[7] += gv[2][1];
                gv_out[2] += gv[0][2]; gv_out[5] += gv[1][2]; gv_out[8] += gv[2][2];
            }
        }
    }
}

/* number of (pixel,face) pairs that survive the bbox test -- the "pairs_active" unit of
 * SURVEY.md section 8(d); used by tests to cross-check the GPU instrumentation kernel. */
long sr_oracle_count_pairs(const float *faces, const sr_params *p)
{
    const int B = p->batch, F = p->nfaces, S = p->size;
    const long npix = (long)S * S;
    const float margin = sqrtf(p->dist_eps * p->sigma);
    long total = 0;
#pragma omp parallel for reduction(+ : total) schedule(static)
    for (long i = 0; i < (long)B * npix; i++) {
        const int bn = (int)(i / npix);
        const long pn = i % npix;
        const int yi = S - 1 - (int)(pn / S), xi = (int)(pn % S);
        const float yp = (float)((2. * yi + 1. - S) / S);
        const float xp = (float)((2. * xi + 1. - S) / S);
        for (int fn = 0; fn < F; fn++)
            total += !outside_bbox(xp, yp, faces + ((long)bn * F + fn) * 9, margin);
    }
    return total;
}
