/*
 * sift_oracle.c — CPU restatement of cv2 SIFT detectAndCompute and of the two image
 * preprocessing calls in front of it.  TEST INFRASTRUCTURE ONLY (see sfm_oracle.h).
 *
 * Reference call sites: sfm.py:40 (cv2.pyrDown), sfm.py:243-244 (cv2.cvtColor BGR2GRAY),
 * sfm.py:246-252 (cv2.xfeatures2d.SIFT_create().detectAndCompute(gray, None)).
 *
 * PARITY STATUS: "parity unpinned".  The algorithm lives in OpenCV (xfeatures2d/features2d
 * sift.cpp, un-vendored, version unpinned, not installable here) and the reference holds no
 * keypoint/descriptor vectors.  This file restates the published algorithm (Lowe 2004 as
 * implemented by OpenCV: constants, loop order, float32 arithmetic, border rules), sequentially,
 * pixel by pixel.  Three library routines OpenCV calls are replaced by fixed, fully specified
 * float programs so that a second implementation can reproduce this file bit for bit:
 *   exp      -> sift_expf   (Cephes single-precision polynomial)
 *   powf(2,) -> sift_expf(e * ln2)
 *   cosf/sinf-> sift_sincos (double Taylor after quadrant reduction, rounded to float)
 * fastAtan2 follows OpenCV's degree-7 odd polynomial.
 */
#include "sfm_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- preprocessing */

/* cv2.cvtColor(img, COLOR_BGR2GRAY) on uint8: fixed point, 14 fractional bits. */
void orc_bgr2gray_u8(const uint8_t* bgr, int64_t w, int64_t h, int64_t stride, uint8_t* gray) {
    for (int64_t y = 0; y < h; ++y)
        for (int64_t x = 0; x < w; ++x) {
            const uint8_t* p = bgr + y * stride + 3 * x;
            gray[y * w + x] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
        }
}

static int reflect101(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - p - 2;
    return p;
}

/* cv2.pyrDown on uint8, `ch` interleaved channels: 5x5 binomial, integer, rounding >> 8. */
void orc_pyrdown_u8(const uint8_t* src, int64_t w, int64_t h, int ch, uint8_t* dst) {
    const int64_t dw = (w + 1) / 2, dh = (h + 1) / 2;
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int64_t y = 0; y < dh; ++y)
        for (int64_t x = 0; x < dw; ++x)
            for (int c = 0; c < ch; ++c) {
                int s = 0;
                for (int i = 0; i < 5; ++i) {
                    const int sy = reflect101((int)(2 * y + i - 2), (int)h);
                    int r = 0;
                    for (int j = 0; j < 5; ++j) r += k[j] * src[((int64_t)sy * w + reflect101((int)(2 * x + j - 2), (int)w)) * ch + c];
                    s += k[i] * r;
                }
                dst[(y * dw + x) * ch + c] = (uint8_t)((s + 128) >> 8);
            }
}

/* ---------------------------------------------------------------- scalar math */

static inline float f_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float sift_expf(float x) {
    if (x < -80.f) x = -80.f;
    if (x > 80.f) x = 80.f;
    const float n = rintf(x * 1.44269504f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    const float y = p * (r * r) + r + 1.0f;
    return y * f_from_bits((uint32_t)((int)n + 127) << 23);
}

static inline void sift_sincos(float xf, float* s, float* c) {   /* xf in radians, |xf| < ~7 */
    const double x = (double)xf;
    const double kq = rint(x * 0.63661977236758134308);
    const double r = (x - kq * 1.57079632679489655800) - kq * 6.123233995736766e-17;
    const double r2 = r * r;
    double sp = -7.6471637318198164759e-13;
    sp = sp * r2 + 1.6059043836821614599e-10;
    sp = sp * r2 - 2.5052108385441718775e-8;
    sp = sp * r2 + 2.7557319223985890653e-6;
    sp = sp * r2 - 1.9841269841269841270e-4;
    sp = sp * r2 + 8.3333333333333333333e-3;
    sp = sp * r2 - 1.6666666666666666667e-1;
    const double sn = r + r * r2 * sp;
    double cp = 4.7794773323873852974e-14;
    cp = cp * r2 - 1.1470745597729724714e-11;
    cp = cp * r2 + 2.0876756987868098979e-9;
    cp = cp * r2 - 2.7557319223985890653e-7;
    cp = cp * r2 + 2.4801587301587301587e-5;
    cp = cp * r2 - 1.3888888888888888889e-3;
    cp = cp * r2 + 4.1666666666666666667e-2;
    const double cs = 1.0 - 0.5 * r2 + r2 * r2 * cp;
    const int q = (int)((long long)kq & 3);
    const double S = q == 0 ? sn : q == 1 ? cs : q == 2 ? -sn : -cs;
    const double Cc = q == 0 ? cs : q == 1 ? -sn : q == 2 ? -cs : sn;
    *s = (float)S; *c = (float)Cc;
}

static inline float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * 57.29577951308232f, p3 = -0.3258083974640975f * 57.29577951308232f,
                p5 = 0.1555786518463281f * 57.29577951308232f, p7 = -0.04432655554792128f * 57.29577951308232f;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else          { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor(float v) { return (int)floorf(v); }

/* ---------------------------------------------------------------- scale space */

typedef struct { int w, h; float* d; } Img;
#define PX(im, r, c) ((im)->d[(size_t)(r) * (im)->w + (c)])

static Img img_new(int w, int h) { Img i = {w, h, (float*)malloc(sizeof(float) * (size_t)w * h)}; return i; }

/* getGaussianKernel(n, sigma, CV_32F), n = cvRound(sigma*4*2+1)|1 (GaussianBlur with Size()). */
int orc_sift_gauss_kernel(double sigma, float* k) {
    const int n = cv_round_d(sigma * 4 * 2 + 1) | 1;
    const double s2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; k[i] = (float)exp(s2 * x * x); sum += k[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
    return n;
}

/* Separable float filter, BORDER_REFLECT_101: rows as a plain dot product in tap order, columns in the
 * symmetric form centre + sum_k ky[k] * (below + above). */
static void gaussian_blur(const Img* src, Img* dst, double sigma) {
    float k[128];
    const int n = orc_sift_gauss_kernel(sigma, k), R = n / 2, w = src->w, h = src->h;
    Img tmp = img_new(w, h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = PX(src, y, reflect101(x - R, w)) * k[0];
            for (int j = 1; j < n; ++j) s += PX(src, y, reflect101(x - R + j, w)) * k[j];
            PX(&tmp, y, x) = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = k[R] * PX(&tmp, y, x);
            for (int j = 1; j <= R; ++j) s += k[R + j] * (PX(&tmp, reflect101(y + j, h), x) + PX(&tmp, reflect101(y - j, h), x));
            PX(dst, y, x) = s;
        }
    free(tmp.d);
}

/* resize(gray_f32, 2x, INTER_LINEAR): horizontal pass then vertical pass, taps clamped at the border. */
static Img upsample2(const uint8_t* gray, int w, int h, int stride) {
    Img o = img_new(2 * w, 2 * h);
    float* row = (float*)malloc(sizeof(float) * 2 * (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int dx = 0; dx < 2 * w; ++dx) {
            float fx = (float)((dx + 0.5) * 0.5 - 0.5);
            int sx = cv_floor(fx); fx -= sx;
            if (sx < 0) { sx = 0; fx = 0; }
            if (sx >= w - 1) { sx = w - 1; fx = 0; }
            const float a = (float)gray[(size_t)y * stride + sx], b = (float)gray[(size_t)y * stride + (sx + 1 < w ? sx + 1 : w - 1)];
            row[(size_t)y * 2 * w + dx] = a * (1.f - fx) + b * fx;
        }
    for (int dy = 0; dy < 2 * h; ++dy) {
        float fy = (float)((dy + 0.5) * 0.5 - 0.5);
        int sy = cv_floor(fy); fy -= sy;
        if (sy < 0) { sy = 0; fy = 0; }
        if (sy >= h - 1) { sy = h - 1; fy = 0; }
        const int sy1 = sy + 1 < h ? sy + 1 : h - 1;
        for (int dx = 0; dx < 2 * w; ++dx)
            PX(&o, dy, dx) = row[(size_t)sy * 2 * w + dx] * (1.f - fy) + row[(size_t)sy1 * 2 * w + dx] * fy;
    }
    free(row);
    return o;
}

typedef struct { float x, y, size, angle, response; int octave; } Kp;

#define SIFT_IMG_BORDER 5
#define SIFT_MAX_INTERP_STEPS 5
#define SIFT_ORI_HIST_BINS 36
#define SIFT_ORI_SIG_FCTR 1.5f
#define SIFT_ORI_RADIUS 4.5f
#define SIFT_ORI_PEAK_RATIO 0.8f
#define SIFT_DESCR_WIDTH 4
#define SIFT_DESCR_HIST_BINS 8
#define SIFT_DESCR_SCL_FCTR 3.f
#define SIFT_DESCR_MAG_THR 0.2f
#define SIFT_INT_DESCR_FCTR 512.f

/* Matx33f::solve(b, DECOMP_LU): the 3x3 specialisation is Cramer's rule. */
static int solve3(const float a[3][3], const float b[3], float x[3]) {
    float d = a[0][0] * (a[1][1] * a[2][2] - a[2][1] * a[1][2]) - a[0][1] * (a[1][0] * a[2][2] - a[2][0] * a[1][2]) +
              a[0][2] * (a[1][0] * a[2][1] - a[2][0] * a[1][1]);
    if (d == 0) { x[0] = x[1] = x[2] = 0; return 0; }
    d = 1 / d;
    x[0] = d * (b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2]) +
                a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
    x[1] = d * (a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) - b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
    x[2] = d * (a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) - a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0]) +
                b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
    return 1;
}

static int adjust_local_extrema(const Img* dog /* octave's DoG stack */, Kp* kpt, int octv, int* layer, int* r, int* c,
                                int nOctaveLayers, float contrastThreshold, float edgeThreshold, float sigma) {
    const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale, cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0;
    for (; i < SIFT_MAX_INTERP_STEPS; ++i) {
        const Img *img = &dog[*layer], *prev = &dog[*layer - 1], *next = &dog[*layer + 1];
        const float dD[3] = {(PX(img, *r, *c + 1) - PX(img, *r, *c - 1)) * deriv_scale, (PX(img, *r + 1, *c) - PX(img, *r - 1, *c)) * deriv_scale,
                             (PX(next, *r, *c) - PX(prev, *r, *c)) * deriv_scale};
        const float v2 = PX(img, *r, *c) * 2;
        const float dxx = (PX(img, *r, *c + 1) + PX(img, *r, *c - 1) - v2) * second_deriv_scale;
        const float dyy = (PX(img, *r + 1, *c) + PX(img, *r - 1, *c) - v2) * second_deriv_scale;
        const float dss = (PX(next, *r, *c) + PX(prev, *r, *c) - v2) * second_deriv_scale;
        const float dxy = (PX(img, *r + 1, *c + 1) - PX(img, *r + 1, *c - 1) - PX(img, *r - 1, *c + 1) + PX(img, *r - 1, *c - 1)) * cross_deriv_scale;
        const float dxs = (PX(next, *r, *c + 1) - PX(next, *r, *c - 1) - PX(prev, *r, *c + 1) + PX(prev, *r, *c - 1)) * cross_deriv_scale;
        const float dys = (PX(next, *r + 1, *c) - PX(next, *r - 1, *c) - PX(prev, *r + 1, *c) + PX(prev, *r - 1, *c)) * cross_deriv_scale;
        const float H[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        float X[3];
        solve3(H, dD, X);
        xi = -X[2]; xr = -X[1]; xc = -X[0];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > (float)(INT32_MAX / 3) || fabsf(xr) > (float)(INT32_MAX / 3) || fabsf(xc) > (float)(INT32_MAX / 3)) return 0;
        *c += cv_round(xc); *r += cv_round(xr); *layer += cv_round(xi);
        if (*layer < 1 || *layer > nOctaveLayers || *c < SIFT_IMG_BORDER || *c >= img->w - SIFT_IMG_BORDER || *r < SIFT_IMG_BORDER ||
            *r >= img->h - SIFT_IMG_BORDER)
            return 0;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return 0;
    {
        const Img *img = &dog[*layer], *prev = &dog[*layer - 1], *next = &dog[*layer + 1];
        const float dD[3] = {(PX(img, *r, *c + 1) - PX(img, *r, *c - 1)) * deriv_scale, (PX(img, *r + 1, *c) - PX(img, *r - 1, *c)) * deriv_scale,
                             (PX(next, *r, *c) - PX(prev, *r, *c)) * deriv_scale};
        const float t = dD[0] * xc + dD[1] * xr + dD[2] * xi;
        const float contr = PX(img, *r, *c) * img_scale + t * 0.5f;
        if (fabsf(contr) * nOctaveLayers < contrastThreshold) return 0;
        const float v2 = PX(img, *r, *c) * 2.f;
        const float dxx = (PX(img, *r, *c + 1) + PX(img, *r, *c - 1) - v2) * second_deriv_scale;
        const float dyy = (PX(img, *r + 1, *c) + PX(img, *r - 1, *c) - v2) * second_deriv_scale;
        const float dxy = (PX(img, *r + 1, *c + 1) - PX(img, *r + 1, *c - 1) - PX(img, *r - 1, *c + 1) + PX(img, *r - 1, *c - 1)) * cross_deriv_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * edgeThreshold >= (edgeThreshold + 1) * (edgeThreshold + 1) * det) return 0;
        kpt->x = (*c + xc) * (1 << octv);
        kpt->y = (*r + xr) * (1 << octv);
        kpt->octave = octv + (*layer << 8) + (cv_round((xi + 0.5f) * 255) << 16);
        kpt->size = sigma * sift_expf(((*layer + xi) / nOctaveLayers) * 0.69314718f) * (1 << octv) * 2;
        kpt->response = fabsf(contr);
    }
    return 1;
}

static float calc_orientation_hist(const Img* img, int px, int py, int radius, float sigma, float* hist, int n) {
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    float temphist[SIFT_ORI_HIST_BINS + 4];
    float* th = temphist + 2;
    for (int i = 0; i < n; ++i) th[i] = 0.f;
    for (int i = -radius; i <= radius; ++i) {
        const int y = py + i;
        if (y <= 0 || y >= img->h - 1) continue;
        for (int j = -radius; j <= radius; ++j) {
            const int x = px + j;
            if (x <= 0 || x >= img->w - 1) continue;
            const float dx = PX(img, y, x + 1) - PX(img, y, x - 1), dy = PX(img, y - 1, x) - PX(img, y + 1, x);
            const float W = sift_expf((i * i + j * j) * expf_scale);
            const float ori = fast_atan2_deg(dy, dx), mag = sqrtf(dx * dx + dy * dy);
            int bin = cv_round((n / 360.f) * ori);
            if (bin >= n) bin -= n;
            if (bin < 0) bin += n;
            th[bin] += W * mag;
        }
    }
    th[-1] = th[n - 1]; th[-2] = th[n - 2]; th[n] = th[0]; th[n + 1] = th[1];
    float maxval = 0;
    for (int i = 0; i < n; ++i) {
        hist[i] = (th[i - 2] + th[i + 2]) * (1.f / 16.f) + (th[i - 1] + th[i + 1]) * (4.f / 16.f) + th[i] * (6.f / 16.f);
        if (i == 0 || hist[i] > maxval) maxval = hist[i];
    }
    return maxval;
}

static void calc_descriptor(const Img* img, float ptx, float pty, float ori, float scl, float* dst) {
    const int d = SIFT_DESCR_WIDTH, n = SIFT_DESCR_HIST_BINS;
    const int px = cv_round(ptx), py = cv_round(pty);
    float cos_t, sin_t;
    sift_sincos(ori * (float)(3.14159265358979323846 / 180), &sin_t, &cos_t);
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = SIFT_DESCR_SCL_FCTR * scl;
    int radius = cv_round(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    const int diag = (int)sqrt((double)img->w * img->w + (double)img->h * img->h);
    if (radius > diag) radius = diag;
    cos_t /= hist_width; sin_t /= hist_width;
    float hist[(SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_HIST_BINS + 2)];
    memset(hist, 0, sizeof hist);
    for (int i = -radius; i <= radius; ++i)
        for (int j = -radius; j <= radius; ++j) {
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < img->h - 1 && c > 0 && c < img->w - 1)) continue;
            const float dx = PX(img, r, c + 1) - PX(img, r, c - 1), dy = PX(img, r - 1, c) - PX(img, r + 1, c);
            const float W = sift_expf((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            float obin = (fast_atan2_deg(dy, dx) - ori) * bins_per_rad;
            const float mag = sqrtf(dx * dx + dy * dy) * W;
            const int r0 = cv_floor(rbin), c0 = cv_floor(cbin);
            int o0 = cv_floor(obin);
            rbin -= r0; cbin -= c0; obin -= o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111, v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
            const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011, v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
            const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
            hist[idx] += v_rco000; hist[idx + 1] += v_rco001;
            hist[idx + (n + 2)] += v_rco010; hist[idx + (n + 3)] += v_rco011;
            hist[idx + (d + 2) * (n + 2)] += v_rco100; hist[idx + (d + 2) * (n + 2) + 1] += v_rco101;
            hist[idx + (d + 3) * (n + 2)] += v_rco110; hist[idx + (d + 3) * (n + 2) + 1] += v_rco111;
        }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            hist[idx] += hist[idx + n];
            hist[idx + 1] += hist[idx + n + 1];
            for (int k = 0; k < n; ++k) dst[(i * d + j) * n + k] = hist[idx + k];
        }
    const int len = d * d * n;
    float nrm2 = 0;
    for (int k = 0; k < len; ++k) nrm2 += dst[k] * dst[k];
    const float thr = sqrtf(nrm2) * SIFT_DESCR_MAG_THR;
    nrm2 = 0;
    for (int k = 0; k < len; ++k) { const float v = dst[k] < thr ? dst[k] : thr; dst[k] = v; nrm2 += v * v; }
    const float s = sqrtf(nrm2);
    nrm2 = SIFT_INT_DESCR_FCTR / (s > FLT_EPSILON ? s : FLT_EPSILON);
    for (int k = 0; k < len; ++k) {   /* saturate_cast<uchar>: round half to even, clamp 0..255 */
        int v = cv_round(dst[k] * nrm2);
        dst[k] = (float)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

/* KeyPointsFilter::removeDuplicatedSorted ordering. */
static int kp_less(const void* a_, const void* b_) {
    const Kp *a = (const Kp*)a_, *b = (const Kp*)b_;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    if (a->size != b->size) return a->size > b->size ? -1 : 1;
    if (a->angle != b->angle) return a->angle < b->angle ? -1 : 1;
    if (a->response != b->response) return a->response > b->response ? -1 : 1;
    if (a->octave != b->octave) return a->octave > b->octave ? -1 : 1;
    return 0;
}

int orc_sift_num_octaves(int w, int h) {   /* of the doubled base image */
    const int m = (2 * w < 2 * h ? 2 * w : 2 * h);
    return cv_round_d(log((double)m) / log(2.) - 2) + 1;
}

/* cv2 SIFT detectAndCompute(gray, None) with nfeatures = 0.  kp: cap x 8 floats
 * {x, y, size, angle, response, octave (int32 bits), class_id = -1 (int32 bits), 0}; desc: cap x 128.
 * pyr_out (optional): receives octave 0's six Gaussian layers (for unit tests).  Returns the keypoint count
 * (may exceed cap; only the first cap are written). */
int64_t orc_sift_detect_and_compute(const uint8_t* gray, int64_t w, int64_t h, int64_t stride, int nOctaveLayers,
                                    double contrastThreshold, double edgeThreshold, double sigma, int64_t cap,
                                    float* kp_out, float* desc_out, float* pyr_out) {
    const int nL = nOctaveLayers, firstOctave = -1;
    /* createInitialImage */
    Img dbl = upsample2(gray, (int)w, (int)h, (int)stride);
    Img base = img_new(dbl.w, dbl.h);
    { const float sf = (float)sigma; float sd = sf * sf - 0.5f * 0.5f * 4; sd = sqrtf(sd > 0.01f ? sd : 0.01f); gaussian_blur(&dbl, &base, (double)sd); }
    free(dbl.d);
    const int nOctaves = orc_sift_num_octaves((int)w, (int)h);
    /* buildGaussianPyramid / buildDoGPyramid */
    double sig[32];
    sig[0] = sigma;
    { const double k = pow(2., 1. / nL);
      for (int i = 1; i < nL + 3; ++i) { const double sp = pow(k, (double)(i - 1)) * sigma, st = sp * k; sig[i] = sqrt(st * st - sp * sp); } }
    Img* G = (Img*)calloc((size_t)nOctaves * (nL + 3), sizeof(Img));
    Img* D = (Img*)calloc((size_t)nOctaves * (nL + 2), sizeof(Img));
    for (int o = 0; o < nOctaves; ++o)
        for (int i = 0; i < nL + 3; ++i) {
            Img* dst = &G[o * (nL + 3) + i];
            if (o == 0 && i == 0) *dst = base;
            else if (i == 0) {
                const Img* src = &G[(o - 1) * (nL + 3) + nL];
                *dst = img_new(src->w / 2, src->h / 2);
                for (int y = 0; y < dst->h; ++y) for (int x = 0; x < dst->w; ++x) PX(dst, y, x) = PX(src, 2 * y, 2 * x);
            } else {
                const Img* src = &G[o * (nL + 3) + i - 1];
                *dst = img_new(src->w, src->h);
                gaussian_blur(src, dst, sig[i]);
            }
        }
    for (int o = 0; o < nOctaves; ++o)
        for (int i = 0; i < nL + 2; ++i) {
            const Img *a = &G[o * (nL + 3) + i], *b = &G[o * (nL + 3) + i + 1];
            Img* dst = &D[o * (nL + 2) + i];
            *dst = img_new(a->w, a->h);
            for (size_t p = 0; p < (size_t)a->w * a->h; ++p) dst->d[p] = b->d[p] - a->d[p];
        }
    if (pyr_out) for (int i = 0; i < nL + 3; ++i) memcpy(pyr_out + (size_t)i * base.w * base.h, G[i].d, sizeof(float) * (size_t)base.w * base.h);

    /* findScaleSpaceExtrema */
    size_t nk = 0, kcap = 1024;
    Kp* kps = (Kp*)malloc(kcap * sizeof(Kp));
    const int threshold = (int)floor(0.5 * contrastThreshold / nL * 255);
    const int n = SIFT_ORI_HIST_BINS;
    for (int o = 0; o < nOctaves; ++o)
        for (int i = 1; i <= nL; ++i) {
            const Img *dog = &D[o * (nL + 2)], *img = &dog[i], *prev = &dog[i - 1], *next = &dog[i + 1];
            for (int r = SIFT_IMG_BORDER; r < img->h - SIFT_IMG_BORDER; ++r)
                for (int c = SIFT_IMG_BORDER; c < img->w - SIFT_IMG_BORDER; ++c) {
                    const float val = PX(img, r, c);
                    if (!(fabsf(val) > threshold)) continue;
                    int is_max = val > 0, is_min = val < 0;
                    for (int dr = -1; dr <= 1 && (is_max || is_min); ++dr)
                        for (int dc = -1; dc <= 1; ++dc) {
                            const float a = PX(prev, r + dr, c + dc), b = PX(next, r + dr, c + dc), m = PX(img, r + dr, c + dc);
                            if (!(val >= a && val >= b && val >= m)) is_max = 0;
                            if (!(val <= a && val <= b && val <= m)) is_min = 0;
                        }
                    if (!is_max && !is_min) continue;
                    Kp kpt; int r1 = r, c1 = c, layer = i;
                    if (!adjust_local_extrema(dog, &kpt, o, &layer, &r1, &c1, nL, (float)contrastThreshold, (float)edgeThreshold, (float)sigma)) continue;
                    const float scl_octv = kpt.size * 0.5f / (1 << o);
                    float hist[SIFT_ORI_HIST_BINS];
                    const float omax = calc_orientation_hist(&G[o * (nL + 3) + layer], c1, r1, cv_round(SIFT_ORI_RADIUS * scl_octv),
                                                             SIFT_ORI_SIG_FCTR * scl_octv, hist, n);
                    const float mag_thr = omax * SIFT_ORI_PEAK_RATIO;
                    for (int j = 0; j < n; ++j) {
                        const int l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
                        if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
                            float bin = j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
                            bin = bin < 0 ? n + bin : bin >= n ? bin - n : bin;
                            kpt.angle = 360.f - (360.f / n) * bin;
                            if (fabsf(kpt.angle - 360.f) < FLT_EPSILON) kpt.angle = 0.f;
                            if (nk == kcap) { kcap *= 2; kps = (Kp*)realloc(kps, kcap * sizeof(Kp)); }
                            kps[nk++] = kpt;
                        }
                    }
                }
        }
    /* removeDuplicatedSorted, then undo the doubling (firstOctave = -1) */
    qsort(kps, nk, sizeof(Kp), kp_less);
    size_t m = 0;
    for (size_t i = 0; i < nk; ++i) {
        if (i && kps[i].x == kps[m - 1].x && kps[i].y == kps[m - 1].y && kps[i].size == kps[m - 1].size && kps[i].angle == kps[m - 1].angle) continue;
        kps[m++] = kps[i];
    }
    nk = m;
    for (size_t i = 0; i < nk; ++i) {
        kps[i].octave = (kps[i].octave & ~255) | ((kps[i].octave + firstOctave) & 255);
        kps[i].x *= 0.5f; kps[i].y *= 0.5f; kps[i].size *= 0.5f;
    }
    /* calcDescriptors */
    for (size_t i = 0; i < nk && (int64_t)i < cap; ++i) {
        const Kp* k = &kps[i];
        int octave = k->octave & 255; const int layer = (k->octave >> 8) & 255;
        octave = octave < 128 ? octave : (-128 | octave);
        const float scale = octave >= 0 ? 1.f / (1 << octave) : (float)(1 << -octave);
        const float size = k->size * scale;
        float angle = 360.f - k->angle;
        if (fabsf(angle - 360.f) < FLT_EPSILON) angle = 0.f;
        if (desc_out) calc_descriptor(&G[(octave - firstOctave) * (nL + 3) + layer], k->x * scale, k->y * scale, angle, size * 0.5f, desc_out + i * 128);
        if (kp_out) {
            float* q = kp_out + i * 8; const int32_t cid = -1;
            q[0] = k->x; q[1] = k->y; q[2] = k->size; q[3] = k->angle; q[4] = k->response;
            memcpy(&q[5], &k->octave, 4); memcpy(&q[6], &cid, 4); q[7] = 0;
        }
    }
    for (int i = 0; i < nOctaves * (nL + 3); ++i) free(G[i].d);
    for (int i = 0; i < nOctaves * (nL + 2); ++i) free(D[i].d);
    free(G); free(D); free(kps);
    return (int64_t)nk;
}
