/*
 * surfel_trace_oracle.c -- CPU restatement (brute force, no acceleration structure) of the
 * differentiable surfel ray tracer EnvGS calls through `diff_surfel_tracing`.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * PARITY STATUS: "parity unpinned".  The CUDA/OptiX sources of diff-surfel-tracing are NOT in
 * /root/reference (empty submodule).  What IS in tree, and is followed here:
 *   - call boundary, argument and output layout : easyvolcap/utils/optix_utils.py:104-119,188-201
 *   - proxy geometry = 3-sigma quad in the tangent plane (uv in [-3,3]^2) : optix_utils.py:39-69 (get_disks)
 *   - per-bounce `mid` layout (16 ch: rayo 0:3, rayd 3:6, dpt 6, acc 7, norm 8:11, aux 11:13, rgb 13:16) : optix_utils.py:30-37
 *   - near plane 0.2 for camera rays : optix_utils.py:212
 *   - SH basis : easyvolcap/utils/sh_utils.py:642-727 ; colour = clamp_min(SH + 0.5, 0) : optix_utils.py:168-169
 *   - un-normalised ray directions ("must be in z_depth") : optix_utils.py:124-127 -> t is in units of |d|
 * The body (ray/plane intersection in the surfel's uv frame, alpha rule, thresholds, front-to-back
 * compositing) follows the 3DGRT / 2DGS published algorithms with the SAME constants as the rasterizer
 * oracle (alpha = min(0.99, o*exp(-(u^2+v^2)/2)), skip alpha < 1/255, stop when T*(1-alpha) < 1e-4).
 * Definitions this project had to choose (DESIGN.md section "tracer semantics"):
 *   - hits are ordered by (t, surfel id); only hits inside the 3-sigma quad count (that is what the triangle
 *     proxy of the reference can report);
 *   - SH is evaluated along the normalised ray direction;
 *   - normals are flipped to face the ray (n.d < 0), world space;
 *   - dpt = sum w*t (not divided by acc), rgb = sum w*c + T*bg, aux = sum w*others;
 *   - start_from_first: t_min = 0.2 (camera rays) else t > 0;
 *   - grads3D.grad (densification signal) = dL/dmeans3D;
 *   - bounces (max_trace_depth > 0): stage k+1 starts at o + d*dpt_k/acc_k along d - 2(d.n)n with n = normalised
 *     accumulated normal, is traced when aux[0] (specular) > specular_threshold and acc_k > 0.5; stage radiance is
 *     blended rgb_k <- (1-s_k)*rgb_k + s_k*rgb_{k+1}; secondary rays are DETACHED (gradients only through stage 0).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEAR_N 0.2f
#define FAR_N 100.0f
#define ALPHA_CAP 0.99f
#define ALPHA_MIN (1.0f / 255.0f)
#define T_EPS 0.0001f
#define UV_MAX 3.0f
#define MID_CH 16

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct {
    int P, R;             /* surfels, rays */
    int D, M;             /* active SH degree, stored coefficients (0 => colors_precomp) */
    int max_trace_depth;
    int start_from_first;
    int has_others;
    int bg_len;
    float scale_modifier;
    float specular_threshold;
} trc_cfg;

typedef struct { float a[3], b[3], n[3], mu[3], su, sv, opa; } surfel_t;

static void make_surfel(const trc_cfg *cfg, int i, const float *means, const float *scales, const float *rots,
                        const float *opac, surfel_t *s)
{
    const float *q = rots + 4 * i;
    float nn = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    float inv = 1.0f / nn;
    float r = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
    s->a[0] = 1.f - 2.f * (y * y + z * z); s->a[1] = 2.f * (x * y + r * z); s->a[2] = 2.f * (x * z - r * y);
    s->b[0] = 2.f * (x * y - r * z); s->b[1] = 1.f - 2.f * (x * x + z * z); s->b[2] = 2.f * (y * z + r * x);
    s->n[0] = 2.f * (x * z + r * y); s->n[1] = 2.f * (y * z - r * x); s->n[2] = 1.f - 2.f * (x * x + y * y);
    s->mu[0] = means[3 * i]; s->mu[1] = means[3 * i + 1]; s->mu[2] = means[3 * i + 2];
    s->su = scales[2 * i] * cfg->scale_modifier; s->sv = scales[2 * i + 1] * cfg->scale_modifier;
    s->opa = opac[i];
}

typedef struct { float t, u, v, G, alpha, denom; } rhit_t;

/* t_min of a traced call: camera rays skip the near 0.2, reflected rays start at 0; start_from_first == 2 is a bounce stage traced as a call
 * of its own (the drop-in module composes max_trace_depth > 0 stage by stage): it starts just off the surface it left, 1e-3, like the
 * in-line bounce stages of trc_forward */
static float first_tmin(int start_from_first) { return start_from_first == 1 ? NEAR_N : (start_from_first == 2 ? 1.0e-3f : 0.0f); }

/* ray / surfel: returns 1 when the hit counts (inside the 3-sigma quad, alpha >= 1/255, t > tmin) */
static int hit_surfel(const surfel_t *s, const float *o, const float *d, float tmin, rhit_t *h)
{
    float denom = s->n[0] * d[0] + s->n[1] * d[1] + s->n[2] * d[2];
    if (denom == 0.0f) return 0;
    float num = s->n[0] * (s->mu[0] - o[0]) + s->n[1] * (s->mu[1] - o[1]) + s->n[2] * (s->mu[2] - o[2]);
    float t = num / denom;
    if (!(t > tmin)) return 0;
    float qx = o[0] + t * d[0] - s->mu[0], qy = o[1] + t * d[1] - s->mu[1], qz = o[2] + t * d[2] - s->mu[2];
    float u = (s->a[0] * qx + s->a[1] * qy + s->a[2] * qz) / s->su;
    float v = (s->b[0] * qx + s->b[1] * qy + s->b[2] * qz) / s->sv;
    if (!(fabsf(u) <= UV_MAX && fabsf(v) <= UV_MAX)) return 0;
    float G = expf(-0.5f * (u * u + v * v));
    float a = s->opa * G;
    float alpha = a < ALPHA_CAP ? a : ALPHA_CAP;
    if (alpha < ALPHA_MIN) return 0;
    h->t = t; h->u = u; h->v = v; h->G = G; h->alpha = alpha; h->denom = denom;
    return 1;
}

static void sh_basis(int D, const float *dir, float basis[16])
{
    float x = dir[0], y = dir[1], z = dir[2];
    basis[0] = SH_C0;
    if (D > 0) {
        basis[1] = -SH_C1 * y; basis[2] = SH_C1 * z; basis[3] = -SH_C1 * x;
        if (D > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            basis[4] = SH_C2[0] * xy; basis[5] = SH_C2[1] * yz; basis[6] = SH_C2[2] * (2.0f * zz - xx - yy);
            basis[7] = SH_C2[3] * xz; basis[8] = SH_C2[4] * (xx - yy);
            if (D > 2) {
                basis[9] = SH_C3[0] * y * (3.0f * xx - yy); basis[10] = SH_C3[1] * xy * z;
                basis[11] = SH_C3[2] * y * (4.0f * zz - xx - yy); basis[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                basis[13] = SH_C3[4] * x * (4.0f * zz - xx - yy); basis[14] = SH_C3[5] * z * (xx - yy);
                basis[15] = SH_C3[6] * x * (xx - 3.0f * yy);
            }
        }
    }
}

/* d basis / d dir (for the SH view-direction gradient) */
static void sh_basis_grad(int D, const float *dir, float gx[16], float gy[16], float gz[16])
{
    float x = dir[0], y = dir[1], z = dir[2];
    for (int k = 0; k < 16; k++) { gx[k] = gy[k] = gz[k] = 0.f; }
    if (D > 0) {
        gy[1] = -SH_C1; gz[2] = SH_C1; gx[3] = -SH_C1;
        if (D > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            gx[4] = SH_C2[0] * y; gy[4] = SH_C2[0] * x;
            gy[5] = SH_C2[1] * z; gz[5] = SH_C2[1] * y;
            gx[6] = SH_C2[2] * -2.f * x; gy[6] = SH_C2[2] * -2.f * y; gz[6] = SH_C2[2] * 4.f * z;
            gx[7] = SH_C2[3] * z; gz[7] = SH_C2[3] * x;
            gx[8] = SH_C2[4] * 2.f * x; gy[8] = SH_C2[4] * -2.f * y;
            if (D > 2) {
                gx[9] = SH_C3[0] * 6.f * xy; gy[9] = SH_C3[0] * 3.f * (xx - yy);
                gx[10] = SH_C3[1] * yz; gy[10] = SH_C3[1] * xz; gz[10] = SH_C3[1] * xy;
                gx[11] = SH_C3[2] * -2.f * xy; gy[11] = SH_C3[2] * (4.f * zz - xx - 3.f * yy); gz[11] = SH_C3[2] * 8.f * yz;
                gx[12] = SH_C3[3] * -6.f * xz; gy[12] = SH_C3[3] * -6.f * yz; gz[12] = SH_C3[3] * 3.f * (2.f * zz - xx - yy);
                gx[13] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); gy[13] = SH_C3[4] * -2.f * xy; gz[13] = SH_C3[4] * 8.f * xz;
                gx[14] = SH_C3[5] * 2.f * xz; gy[14] = SH_C3[5] * -2.f * yz; gz[14] = SH_C3[5] * (xx - yy);
                gx[15] = SH_C3[6] * 3.f * (xx - yy); gy[15] = SH_C3[6] * -6.f * xy;
            }
        }
    }
}

static void surfel_color(const trc_cfg *cfg, int g, const float *shs, const float *colors_precomp, const float basis[16],
                         float col[3], int clampd[3])
{
    if (cfg->M > 0) {
        const float *sh = shs + (size_t)g * cfg->M * 3;
        int nb = (cfg->D + 1) * (cfg->D + 1);
        for (int c = 0; c < 3; c++) {
            float r = 0.f;
            for (int k = 0; k < nb; k++) r += basis[k] * sh[k * 3 + c];
            r += 0.5f;
            clampd[c] = r < 0.f;
            col[c] = r < 0.f ? 0.f : r;
        }
    } else {
        for (int c = 0; c < 3; c++) { col[c] = colors_precomp[3 * g + c]; clampd[c] = 0; }
    }
}

typedef struct { float t; int id; rhit_t h; } ent_t;
static int ent_cmp(const void *a, const void *b)
{
    const ent_t *x = (const ent_t *)a, *y = (const ent_t *)b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}

typedef struct { float rgb[3], dpt, acc, nrm[3], dist, aux[2], T; int nhit; double dist64, distb; } stage_t;     /* dist64 / distb: the distortion shadow, see trc_set_dist_shadow */

/* one stage: trace ray (o,d), composite front to back.  ents is scratch of size P. */
static void trace_stage(const trc_cfg *cfg, const surfel_t *S, const float *shs, const float *colors_precomp,
                        const float *others, const float *bg, const float *o, const float *d, float tmin, ent_t *ents,
                        stage_t *out, double *wet)
{
    int n = 0;
    for (int i = 0; i < cfg->P; i++) {
        rhit_t h;
        if (hit_surfel(&S[i], o, d, tmin, &h)) { ents[n].t = h.t; ents[n].id = i; ents[n].h = h; n++; }
    }
    qsort(ents, n, sizeof(ent_t), ent_cmp);
    float dl = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float dir[3] = {d[0] / dl, d[1] / dl, d[2] / dl};
    float basis[16];
    sh_basis(cfg->D, dir, basis);
    memset(out, 0, sizeof(*out));
    float T = 1.0f, M1 = 0.f, M2 = 0.f;
    double Td = 1.0, M1d = 0.0, M2d = 0.0;
    const double u = 5.9604644775390625e-8;
    for (int k = 0; k < n; k++) {
        const rhit_t *h = &ents[k].h;
        const int g = ents[k].id;
        float test_T = T * (1.0f - h->alpha);
        if (test_T < T_EPS) break;
        float w = h->alpha * T;
        float col[3]; int cl[3];
        surfel_color(cfg, g, shs, colors_precomp, basis, col, cl);
        float tt = h->t > NEAR_N ? h->t : NEAR_N;
        float m = FAR_N / (FAR_N - NEAR_N) * (1.0f - NEAR_N / tt);
        out->dist += (m * m * (1.0f - T) + M2 - 2.0f * m * M1) * w;
        M1 += m * w; M2 += m * m * w;
        {   /* the distortion in double from the float code's own alpha and t, and the a-priori fp32 bound of the moment form (same definitions as
               orc_render_dist64 in surfel_raster_oracle.c; the tracer's wavefront scans add log2(64) levels to the running sums: j + 10) */
            const double wd = (double)h->alpha * Td, A = 1.0 - Td, j = (double)out->nhit;
            const double md = (double)FAR_N / ((double)FAR_N - (double)NEAR_N) * (1.0 - (double)NEAR_N / (double)tt);
            const double E = md * md * A + M2d - 2.0 * md * M1d;
            out->dist64 += E * wd;
            out->distb += wd * u * ((j + 10.0) * (md * md * A + M2d + 2.0 * fabs(md) * M1d) + (j + 8.0) * md * md + 12.0 * fabs(md) * sqrt(A * (E > 0.0 ? E : 0.0)) + (j + 2.0) * fabs(E));
            M1d += md * wd; M2d += md * md * wd;
            Td *= 1.0 - (double)h->alpha;
        }
        for (int c = 0; c < 3; c++) out->rgb[c] += w * col[c];
        out->dpt += w * h->t;
        out->acc += w;
        float sgn = h->denom < 0.0f ? 1.0f : -1.0f;
        for (int c = 0; c < 3; c++) out->nrm[c] += w * sgn * S[g].n[c];
        if (others) { out->aux[0] += w * others[2 * g]; out->aux[1] += w * others[2 * g + 1]; }
        if (wet) {
#pragma omp atomic
            wet[g] += (double)w;
        }
        T = test_T;
        out->nhit++;
    }
    out->T = T;
    for (int c = 0; c < 3; c++) out->rgb[c] += T * (c < cfg->bg_len ? bg[c] : 0.0f);
}

/*
 * Forward.  Outputs: rgb (R,3) dpt (R) acc (R) norm (R,3) dist (R) aux (R,2) mid (R,16*(depth+1)) wet (P) double,
 * final_T (R) (stage-0 transmittance; saved for the backward), nhits (R) int32.
 */
/* Optional outputs of trc_forward (test infrastructure, set before the call, cleared by it): per ray, stage 0's distortion evaluated in double
 * and the a-priori fp32 rounding bound of its moment form -- what the HIP value is asserted against (tests/test_trace_parity.py). */
static double *g_dist64 = NULL, *g_distb = NULL;
void trc_set_dist_shadow(double *dist64, double *bound) { g_dist64 = dist64; g_distb = bound; }

void trc_forward(const trc_cfg *cfg, const float *ray_o, const float *ray_d, const float *means, const float *scales,
                 const float *rots, const float *opac, const float *shs, const float *colors_precomp, const float *others,
                 const float *bg, float *rgb, float *dpt, float *acc, float *norm, float *dist, float *aux, float *mid,
                 double *wet, float *final_T, int32_t *nhits)
{
    const int P = cfg->P, R = cfg->R, ND = cfg->max_trace_depth + 1;
    surfel_t *S = (surfel_t *)malloc(sizeof(surfel_t) * (P ? P : 1));
    for (int i = 0; i < P; i++) make_surfel(cfg, i, means, scales, rots, opac, &S[i]);
    memset(wet, 0, sizeof(double) * P);
    memset(mid, 0, sizeof(float) * (size_t)R * MID_CH * ND);
#pragma omp parallel
    {
        ent_t *ents = (ent_t *)malloc(sizeof(ent_t) * (P ? P : 1));
#pragma omp for schedule(dynamic, 16)
        for (int r = 0; r < R; r++) {
            float o[3] = {ray_o[3 * r], ray_o[3 * r + 1], ray_o[3 * r + 2]};
            float d[3] = {ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]};
            stage_t st[8];
            int ns = 0;
            float tmin = first_tmin(cfg->start_from_first);
            for (int k = 0; k < ND && k < 8; k++) {
                trace_stage(cfg, S, shs, colors_precomp, others, bg, o, d, tmin, ents, &st[k], wet);     /* wet: summed over ALL stages (a surfel blended only by bounce rays is visible too) */
                float *m = mid + ((size_t)r * ND + k) * MID_CH;
                m[0] = o[0]; m[1] = o[1]; m[2] = o[2]; m[3] = d[0]; m[4] = d[1]; m[5] = d[2];
                m[6] = st[k].dpt; m[7] = st[k].acc; m[8] = st[k].nrm[0]; m[9] = st[k].nrm[1]; m[10] = st[k].nrm[2];
                m[11] = st[k].aux[0]; m[12] = st[k].aux[1]; m[13] = st[k].rgb[0]; m[14] = st[k].rgb[1]; m[15] = st[k].rgb[2];
                ns = k + 1;
                if (k + 1 >= ND) break;
                if (!(st[k].aux[0] > cfg->specular_threshold && st[k].acc > 0.5f)) break;
                float nl = sqrtf(st[k].nrm[0] * st[k].nrm[0] + st[k].nrm[1] * st[k].nrm[1] + st[k].nrm[2] * st[k].nrm[2]);
                if (!(nl > 0.0f)) break;
                float nh[3] = {st[k].nrm[0] / nl, st[k].nrm[1] / nl, st[k].nrm[2] / nl};
                float tdep = st[k].dpt / st[k].acc;
                float dn = d[0] * nh[0] + d[1] * nh[1] + d[2] * nh[2];
                for (int c = 0; c < 3; c++) { o[c] = o[c] + d[c] * tdep; }
                for (int c = 0; c < 3; c++) { d[c] = d[c] - 2.0f * dn * nh[c]; }
                tmin = 1e-3f;
            }
            /* blend the stages back to front */
            float col[3] = {st[ns - 1].rgb[0], st[ns - 1].rgb[1], st[ns - 1].rgb[2]};
            for (int k = ns - 2; k >= 0; k--) {
                float s = st[k].aux[0];
                for (int c = 0; c < 3; c++) col[c] = (1.0f - s) * st[k].rgb[c] + s * col[c];
            }
            for (int c = 0; c < 3; c++) rgb[3 * r + c] = col[c];
            dpt[r] = st[0].dpt; acc[r] = st[0].acc; dist[r] = st[0].dist;
            for (int c = 0; c < 3; c++) norm[3 * r + c] = st[0].nrm[c];
            aux[2 * r] = st[0].aux[0]; aux[2 * r + 1] = st[0].aux[1];
            final_T[r] = st[0].T; nhits[r] = st[0].nhit;
            if (g_dist64) g_dist64[r] = st[0].dist64;
            if (g_distb) g_distb[r] = st[0].distb;
        }
        free(ents);
    }
    free(S);
    g_dist64 = NULL; g_distb = NULL;
}

/*
 * Parity audit of one trace stage (test infrastructure for the 1e-4 contract; the counterpart of orc_render_audit).  Replays the
 * stage per ray exactly as trace_stage does (same float code) next to a double-precision shadow computed from the RAW parameters
 * (quaternion -> frame in double, so that frame rounding is part of the measured uncertainty), and flags a ray as FRAGILE when any
 * decided quantity q lies within  K*|q_f32 - q_f64| + m0*|threshold|  of its threshold or when float and double decide differently.
 * Decisions: |u| <= 3, |v| <= 3, alpha >= 1/255, T*(1-alpha) < 1e-4, and (bounce_thr >= 0) the bounce decisions aux0 > thr, acc > 0.5 of
 * the composited stage.  The hit distance t is NOT among them: it is the sort key of the hit lists, i.e. index work, and the GPU evaluates
 * it in this file's operation order without FMA contraction -- t (hence t > tmin and the (t, id) order) is bit-exact by construction, which
 * the tests assert through `tbits`.
 * Outputs: fragile (R) u8; ids / tbits (R, lcap) = surfel id and float bits of t of the composited hits front to back (first nhit[r]
 * valid); nhit (R).  Non-fragile rays must match the GPU's sorted (t, id) list bit for bit and within 1e-4 in value.
 */
#define AUD_K 16.0
#define AUD_M0 4e-6

typedef struct { double a[3], b[3], n[3], mu[3], su, sv, opa; } surfel64_t;

static void make_surfel64(const trc_cfg *cfg, int i, const float *means, const float *scales, const float *rots, const float *opac, surfel64_t *s)
{
    const float *q = rots + 4 * i;
    double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    double inv = 1.0 / sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    double r = q0 * inv, x = q1 * inv, y = q2 * inv, z = q3 * inv;
    s->a[0] = 1. - 2. * (y * y + z * z); s->a[1] = 2. * (x * y + r * z); s->a[2] = 2. * (x * z - r * y);
    s->b[0] = 2. * (x * y - r * z); s->b[1] = 1. - 2. * (x * x + z * z); s->b[2] = 2. * (y * z + r * x);
    s->n[0] = 2. * (x * z + r * y); s->n[1] = 2. * (y * z - r * x); s->n[2] = 1. - 2. * (x * x + y * y);
    s->mu[0] = means[3 * i]; s->mu[1] = means[3 * i + 1]; s->mu[2] = means[3 * i + 2];
    s->su = (double)scales[2 * i] * (double)cfg->scale_modifier; s->sv = (double)scales[2 * i + 1] * (double)cfg->scale_modifier;
    s->opa = opac[i];
}

typedef struct { float t; int id; float alpha; double t64, alpha64; } aent_t;
static int aent_cmp(const void *a, const void *b)
{
    const aent_t *x = (const aent_t *)a, *y = (const aent_t *)b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}

static int near_thr(double q32, double q64, double thr) { return fabs(q64 - thr) <= AUD_K * fabs(q32 - q64) + AUD_M0 * fabs(thr); }

void trc_audit(const trc_cfg *cfg, const float *ray_o, const float *ray_d, const float *means, const float *scales,
               const float *rots, const float *opac, const float *others, const float *shs, float tmin_override, float bounce_thr,
               uint8_t *fragile, int32_t *ids, uint32_t *tbits, int lcap, int32_t *nhit)
{
    /* shs (optional, with cfg->D / cfg->M): the colour clamp clamp_min(SH + 0.5, 0) is a decision too -- a blended colour channel within its
     * own fp32 rounding of zero flips the clamp and with it that hit's whole SH gradient; such rays are flagged like the other near-threshold ones. */
    const int P = cfg->P, R = cfg->R;
    surfel_t *S = (surfel_t *)malloc(sizeof(surfel_t) * (P ? P : 1));
    surfel64_t *S64 = (surfel64_t *)malloc(sizeof(surfel64_t) * (P ? P : 1));
    for (int i = 0; i < P; i++) { make_surfel(cfg, i, means, scales, rots, opac, &S[i]); make_surfel64(cfg, i, means, scales, rots, opac, &S64[i]); }
    const float tmin = tmin_override >= 0.0f ? tmin_override : first_tmin(cfg->start_from_first);
#pragma omp parallel
    {
        aent_t *ents = (aent_t *)malloc(sizeof(aent_t) * (P ? P : 1));
#pragma omp for schedule(dynamic, 16)
        for (int r = 0; r < R; r++) {
            const float *o = ray_o + 3 * r, *d = ray_d + 3 * r;
            const double o64[3] = {o[0], o[1], o[2]}, d64[3] = {d[0], d[1], d[2]};
            int frag = 0, n = 0;
            float basis[16];
            if (shs && cfg->M > 0) {
                const float il = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                const float dn[3] = {d[0] * il, d[1] * il, d[2] * il};
                for (int k = 0; k < 16; k++) basis[k] = 0.f;
                sh_basis(cfg->D, dn, basis);
            }
            for (int i = 0; i < P; i++) {
                rhit_t h;
                const int ok = hit_surfel(&S[i], o, d, tmin, &h);
                /* the float quantities without the early exits */
                const surfel_t *f = &S[i];
                const float denf = f->n[0] * d[0] + f->n[1] * d[1] + f->n[2] * d[2];
                if (denf == 0.0f) continue;
                const float tf = (f->n[0] * (f->mu[0] - o[0]) + f->n[1] * (f->mu[1] - o[1]) + f->n[2] * (f->mu[2] - o[2])) / denf;
                if (!(tf > tmin)) continue;                 /* t is exact by construction in both implementations: not a fragile decision */
                const float qxf = o[0] + tf * d[0] - f->mu[0], qyf = o[1] + tf * d[1] - f->mu[1], qzf = o[2] + tf * d[2] - f->mu[2];
                const float uf = (f->a[0] * qxf + f->a[1] * qyf + f->a[2] * qzf) / f->su;
                const float vf = (f->b[0] * qxf + f->b[1] * qyf + f->b[2] * qzf) / f->sv;
                const float af0 = f->opa * expf(-0.5f * (uf * uf + vf * vf));
                const float af = af0 < ALPHA_CAP ? af0 : ALPHA_CAP;
                /* the double shadow, from the raw parameters */
                const surfel64_t *s = &S64[i];
                const double den = s->n[0] * d64[0] + s->n[1] * d64[1] + s->n[2] * d64[2];
                double t64 = tf, u64 = uf, v64 = vf, a64 = af;
                if (den != 0.0) {
                    t64 = (s->n[0] * (s->mu[0] - o64[0]) + s->n[1] * (s->mu[1] - o64[1]) + s->n[2] * (s->mu[2] - o64[2])) / den;
                    const double qx = o64[0] + t64 * d64[0] - s->mu[0], qy = o64[1] + t64 * d64[1] - s->mu[1], qz = o64[2] + t64 * d64[2] - s->mu[2];
                    u64 = (s->a[0] * qx + s->a[1] * qy + s->a[2] * qz) / s->su;
                    v64 = (s->b[0] * qx + s->b[1] * qy + s->b[2] * qz) / s->sv;
                    const double a = s->opa * exp(-0.5 * (u64 * u64 + v64 * v64));
                    a64 = a < (double)ALPHA_CAP ? a : (double)ALPHA_CAP;
                }
                const int ok64 = fabs(u64) <= (double)UV_MAX && fabs(v64) <= (double)UV_MAX && a64 >= (double)ALPHA_MIN;
                if (ok != ok64) frag = 1;
                /* only surfels that could make a difference: roughly inside the quad, roughly visible */
                if (fabs(u64) <= UV_MAX + 0.5 && fabs(v64) <= UV_MAX + 0.5 && a64 >= 0.5 * (double)ALPHA_MIN) {
                    if (near_thr(fabsf(uf), fabs(u64), (double)UV_MAX)) frag = 1;
                    if (near_thr(fabsf(vf), fabs(v64), (double)UV_MAX)) frag = 1;
                    if (near_thr(af, a64, (double)ALPHA_MIN)) frag = 1;
                }
                if (ok) { ents[n].t = h.t; ents[n].id = i; ents[n].alpha = h.alpha; ents[n].t64 = t64; ents[n].alpha64 = a64; n++; }
            }
            qsort(ents, n, sizeof(aent_t), aent_cmp);
            float T = 1.0f, acc = 0.f, aux0 = 0.f;
            double T64 = 1.0, acc64 = 0.0, aux064 = 0.0;
            int nh = 0;
            for (int k = 0; k < n; k++) {
                const float test_T = T * (1.0f - ents[k].alpha);
                const double test_T64 = T64 * (1.0 - ents[k].alpha64);
                if (near_thr(test_T, test_T64, (double)T_EPS) || ((test_T < T_EPS) != (test_T64 < (double)T_EPS))) frag = 1;
                if (test_T < T_EPS) break;
                const float w = ents[k].alpha * T;
                const double w64 = ents[k].alpha64 * T64;
                acc += w; acc64 += w64;
                if (others) { aux0 += w * others[2 * ents[k].id]; aux064 += w64 * (double)others[2 * ents[k].id]; }
                if (shs && cfg->M > 0) {
                    const float *sh = shs + (size_t)ents[k].id * cfg->M * 3;
                    const int nbas = (cfg->D + 1) * (cfg->D + 1);
                    for (int c = 0; c < 3; c++) {
                        float rr = 0.f, mag = 0.5f;
                        for (int kk = 0; kk < nbas; kk++) { rr += basis[kk] * sh[kk * 3 + c]; mag += fabsf(basis[kk] * sh[kk * 3 + c]); }
                        rr += 0.5f;
                        /* rounding of a (nbas + 1)-term fp32 sum whose terms carry a few ulp themselves (basis polynomials, summation order) */
                        if (fabsf(rr) <= (float)AUD_K * 8.0f * 1.1920929e-7f * mag) frag = 1;
                    }
                }
                if (nh < lcap) { ids[(size_t)r * lcap + nh] = ents[k].id; union { float f; uint32_t u; } cv; cv.f = ents[k].t; tbits[(size_t)r * lcap + nh] = cv.u; }
                nh++;
                T = test_T; T64 = test_T64;
            }
            if (bounce_thr >= 0.0f) {
                if (near_thr(aux0, aux064, (double)bounce_thr) || near_thr(acc, acc64, 0.5)) frag = 1;
            }
            fragile[r] = (uint8_t)frag;
            nhit[r] = nh;
        }
        free(ents);
    }
    free(S); free(S64);
}

/*
 * Backward of stage 0 (max_trace_depth == 0 semantics; secondary rays are detached).
 * Upstream: dL_drgb (R,3) dL_ddpt (R) dL_dacc (R) dL_dnorm (R,3) dL_daux (R,2).   (dist carries no gradient here.)
 * Outputs (double, zeroed here): dmeans (P,3) dscales (P,2) drots (P,4) dopac (P) dshs (P,M,3) | dcolors (P,3),
 * dothers (P,2), dray_o (R,3), dray_d (R,3).
 * The 0.99 alpha cap is treated as identity in the gradient, like the rasterizer.
 */
void trc_backward(const trc_cfg *cfg, const float *ray_o, const float *ray_d, const float *means, const float *scales,
                  const float *rots, const float *opac, const float *shs, const float *colors_precomp, const float *others,
                  const float *bg, const float *dL_drgb, const float *dL_ddpt, const float *dL_dacc, const float *dL_dnorm,
                  const float *dL_daux, double *dmeans, double *dscales, double *drots, double *dopac, double *dshs,
                  double *dcolors, double *dothers, double *dray_o, double *dray_d,
                  double *c_means, double *c_scales, double *c_rots, double *c_opac, double *c_color, double *c_others,
                  double *c_ray_o, double *c_ray_d,
                  double *u_c_means, double *u_c_scales, double *u_c_rots, double *u_c_opac, double *u_c_color, double *u_c_others,
                  double *u_c_ray_o, double *u_c_ray_d)
{
    /* u_* (optional, with c_*): the oracle's own fp32 UNCERTAINTY of every gradient element -- the same noise-scale accumulation with each
     * hit weighted by rho = |w32 - w64| / w64 + |alpha32 - alpha64| / alpha64 + (|u32 - u64| + |v32 - v64|) / (|u| + |v| + 0.01), the relative
     * distance between this file's float evaluation of the hit and a double evaluation from the raw parameters (the blend weight of the
     * k-th hit carries the rounding of k transmittance factors: ~1e-4 after a hundred hits of a fog, in any fp32 implementation). */
    /* c_* (optional, all or none; c_color is (P,M,3) or (P,3)): the NOISE SCALE of every gradient element -- the same accumulation with
     * every hit's dL/dalpha replaced by the sum of the magnitudes it is a difference of (|g| * (T |value| + (|final| + |prefix|) / (1 - alpha)))
     * and every coefficient by its absolute value.  An fp32 implementation that forms "suffix = final - prefix" carries an error of a few
     * ulp of THOSE magnitudes into each hit's gradient, so this -- not the (possibly tiny) gradient itself -- is what its error is
     * measured against (tests/util.py: condition-aware floor).  The oracle itself keeps all per-ray sums in double. */
    const int P = cfg->P, R = cfg->R, M = cfg->M;
    surfel_t *S = (surfel_t *)malloc(sizeof(surfel_t) * (P ? P : 1));
    for (int i = 0; i < P; i++) make_surfel(cfg, i, means, scales, rots, opac, &S[i]);
    double *dA = (double *)calloc((size_t)3 * P + 1, sizeof(double)), *dB = (double *)calloc((size_t)3 * P + 1, sizeof(double)),
           *dN = (double *)calloc((size_t)3 * P + 1, sizeof(double));
    memset(dmeans, 0, sizeof(double) * 3 * P); memset(dscales, 0, sizeof(double) * 2 * P);
    memset(drots, 0, sizeof(double) * 4 * P); memset(dopac, 0, sizeof(double) * P);
    if (M > 0) memset(dshs, 0, sizeof(double) * (size_t)P * M * 3); else memset(dcolors, 0, sizeof(double) * 3 * P);
    if (others) memset(dothers, 0, sizeof(double) * 2 * P);
    memset(dray_o, 0, sizeof(double) * 3 * R); memset(dray_d, 0, sizeof(double) * 3 * R);
    double *cA_ = NULL, *cB_ = NULL, *cN_ = NULL;
    if (c_means) {
        memset(c_means, 0, sizeof(double) * 3 * P); memset(c_scales, 0, sizeof(double) * 2 * P); memset(c_rots, 0, sizeof(double) * 4 * P);
        memset(c_opac, 0, sizeof(double) * P); memset(c_color, 0, sizeof(double) * (size_t)P * (M > 0 ? M * 3 : 3));
        if (others) memset(c_others, 0, sizeof(double) * 2 * P);
        memset(c_ray_o, 0, sizeof(double) * 3 * R); memset(c_ray_d, 0, sizeof(double) * 3 * R);
        cA_ = (double *)calloc((size_t)3 * P + 1, sizeof(double)); cB_ = (double *)calloc((size_t)3 * P + 1, sizeof(double));
        cN_ = (double *)calloc((size_t)3 * P + 1, sizeof(double));
    }
    double *u_cA_ = NULL, *u_cB_ = NULL, *u_cN_ = NULL;
    surfel64_t *S64 = NULL;
    if (u_c_means) {
        memset(u_c_means, 0, sizeof(double) * 3 * P); memset(u_c_scales, 0, sizeof(double) * 2 * P); memset(u_c_rots, 0, sizeof(double) * 4 * P);
        memset(u_c_opac, 0, sizeof(double) * P); memset(u_c_color, 0, sizeof(double) * (size_t)P * (M > 0 ? M * 3 : 3));
        if (others) memset(u_c_others, 0, sizeof(double) * 2 * P);
        memset(u_c_ray_o, 0, sizeof(double) * 3 * R); memset(u_c_ray_d, 0, sizeof(double) * 3 * R);
        u_cA_ = (double *)calloc((size_t)3 * P + 1, sizeof(double)); u_cB_ = (double *)calloc((size_t)3 * P + 1, sizeof(double));
        u_cN_ = (double *)calloc((size_t)3 * P + 1, sizeof(double));
        S64 = (surfel64_t *)malloc(sizeof(surfel64_t) * (P ? P : 1));
        for (int i = 0; i < P; i++) make_surfel64(cfg, i, means, scales, rots, opac, &S64[i]);
    }
#define ACC(arr, idx, v) do { double v__ = (double)(v); _Pragma("omp atomic") arr[idx] += v__; } while (0)
#define CACC(arr, idx, v) do { if (c_means) { double v__ = fabs((double)(v)); _Pragma("omp atomic") arr[idx] += v__; \
                               if (u_c_means) { double w__ = rho * v__; _Pragma("omp atomic") u_##arr[idx] += w__; } } } while (0)
#pragma omp parallel
    {
        ent_t *ents = (ent_t *)malloc(sizeof(ent_t) * (P ? P : 1));
#pragma omp for schedule(dynamic, 16)
        for (int r = 0; r < R; r++) {
            const float o[3] = {ray_o[3 * r], ray_o[3 * r + 1], ray_o[3 * r + 2]};
            const float d[3] = {ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]};
            const float tmin = first_tmin(cfg->start_from_first);
            stage_t fin;
            trace_stage(cfg, S, shs, colors_precomp, others, bg, o, d, tmin, ents, &fin, NULL);   /* final sums; ents sorted */
            int n = 0;
            for (int i = 0; i < P; i++) { rhit_t h; if (hit_surfel(&S[i], o, d, tmin, &h)) { ents[n].t = h.t; ents[n].id = i; ents[n].h = h; n++; } }
            qsort(ents, n, sizeof(ent_t), ent_cmp);
            const float dl2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], dl = sqrtf(dl2);
            const float dir[3] = {d[0] / dl, d[1] / dl, d[2] / dl};
            float basis[16], bgx[16], bgy[16], bgz[16];
            sh_basis(cfg->D, dir, basis);
            sh_basis_grad(cfg->D, dir, bgx, bgy, bgz);
            const float gR[3] = {dL_drgb[3 * r], dL_drgb[3 * r + 1], dL_drgb[3 * r + 2]};
            const float gD = dL_ddpt[r], gA = dL_dacc[r];
            const float gN[3] = {dL_dnorm[3 * r], dL_dnorm[3 * r + 1], dL_dnorm[3 * r + 2]};
            const float gX[2] = {dL_daux[2 * r], dL_daux[2 * r + 1]};
            /* running prefix sums (through hit k) of the composited quantities -- in double, as are the totals they are subtracted from
             * (re-accumulated here: the forward's float totals carry their own rounding), so that the oracle's suffix terms are exact */
            float T = 1.0f;
            double crgb[3] = {0, 0, 0}, cD = 0., cA = 0., cN[3] = {0, 0, 0}, cX[2] = {0, 0};
            double frgb[3] = {0, 0, 0}, fD = 0., fA = 0., fN[3] = {0, 0, 0}, fX[2] = {0, 0};
            {
                float Tt = 1.0f;
                for (int k = 0; k < n; k++) {
                    const rhit_t *h = &ents[k].h;
                    const int g = ents[k].id;
                    const float tt = Tt * (1.0f - h->alpha);
                    if (tt < T_EPS) break;
                    const double w = (double)(h->alpha * Tt);
                    float col[3]; int cl[3];
                    surfel_color(cfg, g, shs, colors_precomp, basis, col, cl);
                    const double sg = h->denom < 0.0f ? 1.0 : -1.0;
                    for (int c = 0; c < 3; c++) { frgb[c] += w * col[c]; fN[c] += w * sg * S[g].n[c]; }
                    fD += w * h->t; fA += w;
                    if (others) { fX[0] += w * others[2 * g]; fX[1] += w * others[2 * g + 1]; }
                    Tt = tt;
                }
            }
            float bgdot = 0.f;
            for (int c = 0; c < 3; c++) bgdot += (c < cfg->bg_len ? bg[c] : 0.0f) * gR[c];
            double ddir[3] = {0, 0, 0}, dO[3] = {0, 0, 0}, dD[3] = {0, 0, 0};
            double nddir[3] = {0, 0, 0}, nO[3] = {0, 0, 0}, nD[3] = {0, 0, 0};
            double uddir[3] = {0, 0, 0}, uO[3] = {0, 0, 0}, uD[3] = {0, 0, 0}, T64 = 1.0, rho = 0.0;
            for (int k = 0; k < n; k++) {
                const rhit_t *h = &ents[k].h;
                const int g = ents[k].id;
                const surfel_t *s = &S[g];
                const float alpha = h->alpha;
                const float test_T = T * (1.0f - alpha);
                if (test_T < T_EPS) break;
                const float w = alpha * T;
                if (S64) {
                    const surfel64_t *z = &S64[g];
                    const double den = z->n[0] * d[0] + z->n[1] * d[1] + z->n[2] * d[2];
                    const double t64 = (z->n[0] * (z->mu[0] - o[0]) + z->n[1] * (z->mu[1] - o[1]) + z->n[2] * (z->mu[2] - o[2])) / den;
                    const double qx64 = o[0] + t64 * d[0] - z->mu[0], qy64 = o[1] + t64 * d[1] - z->mu[1], qz64 = o[2] + t64 * d[2] - z->mu[2];
                    const double u64 = (z->a[0] * qx64 + z->a[1] * qy64 + z->a[2] * qz64) / z->su, v64 = (z->b[0] * qx64 + z->b[1] * qy64 + z->b[2] * qz64) / z->sv;
                    double a64 = z->opa * exp(-0.5 * (u64 * u64 + v64 * v64));
                    if (a64 > (double)ALPHA_CAP) a64 = (double)ALPHA_CAP;
                    const double w64 = a64 * T64;
                    rho = fabs((double)w - w64) / (w64 > 1e-30 ? w64 : 1e-30) + fabs((double)alpha - a64) / a64 +
                          (fabs((double)h->u - u64) + fabs((double)h->v - v64)) / (fabs(u64) + fabs(v64) + 0.01);
                    T64 *= (1.0 - a64);
                }
                float col[3]; int cl[3];
                surfel_color(cfg, g, shs, colors_precomp, basis, col, cl);
                const float sgn = h->denom < 0.0f ? 1.0f : -1.0f;
                const float nf[3] = {sgn * s->n[0], sgn * s->n[1], sgn * s->n[2]};
                const float ox0 = others ? others[2 * g] : 0.f, ox1 = others ? others[2 * g + 1] : 0.f;
                /* prefix through k */
                for (int c = 0; c < 3; c++) crgb[c] += w * col[c];
                cD += w * h->t; cA += w;
                for (int c = 0; c < 3; c++) cN[c] += w * nf[c];
                cX[0] += w * ox0; cX[1] += w * ox1;
                /* dL/dalpha_k = T_k * value_k - (suffix after k) / (1 - alpha_k);  nLa = the magnitudes this is a difference of */
                const double inv1m = 1.0 / (1.0 - (double)alpha);
                double dLa = 0., nLa = 0.;
                for (int c = 0; c < 3; c++) {
                    dLa += gR[c] * ((double)T * col[c] - (frgb[c] - crgb[c]) * inv1m);
                    nLa += fabs(gR[c]) * ((double)T * fabs(col[c]) + (fabs(frgb[c]) + fabs(crgb[c])) * inv1m);
                }
                dLa += gD * ((double)T * h->t - (fD - cD) * inv1m);                 nLa += fabs(gD) * ((double)T * fabs(h->t) + (fabs(fD) + fabs(cD)) * inv1m);
                dLa += gA * ((double)T - (fA - cA) * inv1m);                        nLa += fabs(gA) * ((double)T + (fabs(fA) + fabs(cA)) * inv1m);
                for (int c = 0; c < 3; c++) {
                    dLa += gN[c] * ((double)T * nf[c] - (fN[c] - cN[c]) * inv1m);
                    nLa += fabs(gN[c]) * ((double)T * fabs(nf[c]) + (fabs(fN[c]) + fabs(cN[c])) * inv1m);
                }
                dLa += gX[0] * ((double)T * ox0 - (fX[0] - cX[0]) * inv1m) + gX[1] * ((double)T * ox1 - (fX[1] - cX[1]) * inv1m);
                nLa += fabs(gX[0]) * ((double)T * fabs(ox0) + (fabs(fX[0]) + fabs(cX[0])) * inv1m) + fabs(gX[1]) * ((double)T * fabs(ox1) + (fabs(fX[1]) + fabs(cX[1])) * inv1m);
                dLa += -((double)fin.T * inv1m) * bgdot;                            nLa += fabs((double)fin.T * inv1m * bgdot);
                /* direct terms */
                float dcol[3];
                for (int c = 0; c < 3; c++) dcol[c] = cl[c] ? 0.f : w * gR[c];
                if (M > 0) {
                    int nb = (cfg->D + 1) * (cfg->D + 1);
                    const float *sh = shs + (size_t)g * M * 3;
                    for (int kk = 0; kk < nb; kk++)
                        for (int c = 0; c < 3; c++) {
                            ACC(dshs, ((size_t)g * M + kk) * 3 + c, basis[kk] * dcol[c]);
                            CACC(c_color, ((size_t)g * M + kk) * 3 + c, basis[kk] * dcol[c]);
                            ddir[0] += (double)(bgx[kk] * sh[kk * 3 + c] * dcol[c]); nddir[0] += fabs((double)(bgx[kk] * sh[kk * 3 + c] * dcol[c])); uddir[0] += rho * fabs((double)(bgx[kk] * sh[kk * 3 + c] * dcol[c]));
                            ddir[1] += (double)(bgy[kk] * sh[kk * 3 + c] * dcol[c]); nddir[1] += fabs((double)(bgy[kk] * sh[kk * 3 + c] * dcol[c])); uddir[1] += rho * fabs((double)(bgy[kk] * sh[kk * 3 + c] * dcol[c]));
                            ddir[2] += (double)(bgz[kk] * sh[kk * 3 + c] * dcol[c]); nddir[2] += fabs((double)(bgz[kk] * sh[kk * 3 + c] * dcol[c])); uddir[2] += rho * fabs((double)(bgz[kk] * sh[kk * 3 + c] * dcol[c]));
                        }
                } else {
                    for (int c = 0; c < 3; c++) { ACC(dcolors, 3 * (size_t)g + c, dcol[c]); CACC(c_color, 3 * (size_t)g + c, dcol[c]); }
                }
                if (others) {
                    ACC(dothers, 2 * (size_t)g, w * gX[0]); ACC(dothers, 2 * (size_t)g + 1, w * gX[1]);
                    CACC(c_others, 2 * (size_t)g, w * gX[0]); CACC(c_others, 2 * (size_t)g + 1, w * gX[1]);
                }
                for (int c = 0; c < 3; c++) { ACC(dN, 3 * (size_t)g + c, w * sgn * gN[c]); CACC(cN_, 3 * (size_t)g + c, w * gN[c]); }
                const double dLt = (double)w * gD;
                /* alpha = opa * G ; G = exp(-(u^2+v^2)/2) */
                ACC(dopac, g, h->G * dLa); CACC(c_opac, g, h->G * nLa);
                const double dLG = s->opa * dLa, nLG = s->opa * nLa;
                const double dLu = dLG * (-h->G * h->u), dLv = dLG * (-h->G * h->v);
                const double nLu = nLG * fabs(h->G * h->u), nLv = nLG * fabs(h->G * h->v);
                const float qx = o[0] + h->t * d[0] - s->mu[0], qy = o[1] + h->t * d[1] - s->mu[1], qz = o[2] + h->t * d[2] - s->mu[2];
                const float q[3] = {qx, qy, qz};
                const double cu = dLu / s->su, cv = dLv / s->sv, ncu = nLu / s->su, ncv = nLv / s->sv;
                double dq[3], nq[3];
                for (int c = 0; c < 3; c++) { dq[c] = cu * s->a[c] + cv * s->b[c]; nq[c] = ncu * fabs(s->a[c]) + ncv * fabs(s->b[c]); }
                for (int c = 0; c < 3; c++) {
                    ACC(dA, 3 * (size_t)g + c, cu * q[c]); ACC(dB, 3 * (size_t)g + c, cv * q[c]);
                    CACC(cA_, 3 * (size_t)g + c, ncu * q[c]); CACC(cB_, 3 * (size_t)g + c, ncv * q[c]);
                }
                ACC(dscales, 2 * (size_t)g, -dLu * h->u / s->su * cfg->scale_modifier);
                ACC(dscales, 2 * (size_t)g + 1, -dLv * h->v / s->sv * cfg->scale_modifier);
                CACC(c_scales, 2 * (size_t)g, nLu * h->u / s->su * cfg->scale_modifier);
                CACC(c_scales, 2 * (size_t)g + 1, nLv * h->v / s->sv * cfg->scale_modifier);
                /* q = o + t d - mu */
                const double dLt_tot = dLt + dq[0] * d[0] + dq[1] * d[1] + dq[2] * d[2];
                const double nLt_tot = fabs(dLt) + nq[0] * fabs(d[0]) + nq[1] * fabs(d[1]) + nq[2] * fabs(d[2]);
                const double k_t = dLt_tot / h->denom, nk_t = nLt_tot / fabs(h->denom);
                for (int c = 0; c < 3; c++) {
                    ACC(dmeans, 3 * (size_t)g + c, -dq[c] + k_t * s->n[c]);
                    CACC(c_means, 3 * (size_t)g + c, nq[c] + nk_t * fabs(s->n[c]));
                    dO[c] += (double)(dq[c] - k_t * s->n[c]);                   nO[c] += nq[c] + nk_t * fabs(s->n[c]);                    uO[c] += rho * (nq[c] + nk_t * fabs(s->n[c]));
                    dD[c] += (double)(h->t * dq[c] - k_t * h->t * s->n[c]);     nD[c] += fabs(h->t) * (nq[c] + nk_t * fabs(s->n[c]));     uD[c] += rho * fabs(h->t) * (nq[c] + nk_t * fabs(s->n[c]));
                    ACC(dN, 3 * (size_t)g + c, -k_t * q[c]); CACC(cN_, 3 * (size_t)g + c, nk_t * q[c]);
                }
                T = test_T;
            }
            /* SH direction gradient back through d/|d| */
            {
                double dd0 = ddir[0], dd1 = ddir[1], dd2 = ddir[2];
                double inv3 = 1.0 / ((double)dl2 * (double)dl);
                dD[0] += ((dl2 - d[0] * d[0]) * dd0 - d[1] * d[0] * dd1 - d[2] * d[0] * dd2) * inv3;
                dD[1] += (-d[0] * d[1] * dd0 + (dl2 - d[1] * d[1]) * dd1 - d[2] * d[1] * dd2) * inv3;
                dD[2] += (-d[0] * d[2] * dd0 - d[1] * d[2] * dd1 + (dl2 - d[2] * d[2]) * dd2) * inv3;
            }
            for (int c = 0; c < 3; c++) { dray_o[3 * r + c] = dO[c]; dray_d[3 * r + c] = dD[c]; }
            if (c_means) {
                double inv3 = 1.0 / ((double)dl2 * (double)dl);
                for (int c = 0; c < 3; c++) {
                    double sd = 0.;
                    for (int e = 0; e < 3; e++) sd += fabs((e == c ? (double)dl2 : 0.0) - (double)d[c] * d[e]) * nddir[e];
                    c_ray_o[3 * r + c] = nO[c]; c_ray_d[3 * r + c] = nD[c] + sd * inv3;
                    if (u_c_means) {
                        double su_ = 0.;
                        for (int e = 0; e < 3; e++) su_ += fabs((e == c ? (double)dl2 : 0.0) - (double)d[c] * d[e]) * uddir[e];
                        u_c_ray_o[3 * r + c] = uO[c]; u_c_ray_d[3 * r + c] = uD[c] + su_ * inv3;
                    }
                }
            }
        }
        free(ents);
    }
#undef ACC
#undef CACC
    /* columns a,b,n of R -> unit quaternion gradient (no projection through the normalisation) */
    for (int i = 0; i < P; i++) {
        const float *q = rots + 4 * i;
        float nn = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        double r = q[0] / nn, x = q[1] / nn, y = q[2] / nn, z = q[3] / nn;
        double V[3][3];
        for (int rr = 0; rr < 3; rr++) { V[rr][0] = dA[3 * i + rr]; V[rr][1] = dB[3 * i + rr]; V[rr][2] = dN[3 * i + rr]; }
        drots[4 * i + 0] = 2. * (x * (V[2][1] - V[1][2]) + y * (V[0][2] - V[2][0]) + z * (V[1][0] - V[0][1]));
        drots[4 * i + 1] = 2. * (-2. * x * (V[1][1] + V[2][2]) + y * (V[1][0] + V[0][1]) + z * (V[2][0] + V[0][2]) + r * (V[2][1] - V[1][2]));
        drots[4 * i + 2] = 2. * (x * (V[1][0] + V[0][1]) - 2. * y * (V[0][0] + V[2][2]) + z * (V[2][1] + V[1][2]) + r * (V[0][2] - V[2][0]));
        drots[4 * i + 3] = 2. * (x * (V[2][0] + V[0][2]) + y * (V[2][1] + V[1][2]) - 2. * z * (V[0][0] + V[1][1]) + r * (V[1][0] - V[0][1]));
        if (c_means) {
            double W[3][3], ar = fabs(r), ax = fabs(x), ay = fabs(y), az = fabs(z);
            for (int rr = 0; rr < 3; rr++) { W[rr][0] = cA_[3 * i + rr]; W[rr][1] = cB_[3 * i + rr]; W[rr][2] = cN_[3 * i + rr]; }
            c_rots[4 * i + 0] = 2. * (ax * (W[2][1] + W[1][2]) + ay * (W[0][2] + W[2][0]) + az * (W[1][0] + W[0][1]));
            c_rots[4 * i + 1] = 2. * (2. * ax * (W[1][1] + W[2][2]) + ay * (W[1][0] + W[0][1]) + az * (W[2][0] + W[0][2]) + ar * (W[2][1] + W[1][2]));
            c_rots[4 * i + 2] = 2. * (ax * (W[1][0] + W[0][1]) + 2. * ay * (W[0][0] + W[2][2]) + az * (W[2][1] + W[1][2]) + ar * (W[0][2] + W[2][0]));
            c_rots[4 * i + 3] = 2. * (ax * (W[2][0] + W[0][2]) + ay * (W[2][1] + W[1][2]) + 2. * az * (W[0][0] + W[1][1]) + ar * (W[1][0] + W[0][1]));
            if (u_c_means) {
                for (int rr = 0; rr < 3; rr++) { W[rr][0] = u_cA_[3 * i + rr]; W[rr][1] = u_cB_[3 * i + rr]; W[rr][2] = u_cN_[3 * i + rr]; }
                u_c_rots[4 * i + 0] = 2. * (ax * (W[2][1] + W[1][2]) + ay * (W[0][2] + W[2][0]) + az * (W[1][0] + W[0][1]));
                u_c_rots[4 * i + 1] = 2. * (2. * ax * (W[1][1] + W[2][2]) + ay * (W[1][0] + W[0][1]) + az * (W[2][0] + W[0][2]) + ar * (W[2][1] + W[1][2]));
                u_c_rots[4 * i + 2] = 2. * (ax * (W[1][0] + W[0][1]) + 2. * ay * (W[0][0] + W[2][2]) + az * (W[2][1] + W[1][2]) + ar * (W[0][2] + W[2][0]));
                u_c_rots[4 * i + 3] = 2. * (ax * (W[2][0] + W[0][2]) + ay * (W[2][1] + W[1][2]) + 2. * az * (W[0][0] + W[1][1]) + ar * (W[1][0] + W[0][1]));
            }
        }
    }
    free(dA); free(dB); free(dN); free(S);
    if (cA_) { free(cA_); free(cB_); free(cN_); }
    if (u_cA_) { free(u_cA_); free(u_cB_); free(u_cN_); free(S64); }
}
