/* mcalf_oracle.c -- plain-C restatement of the MC-ALF numpy likelihood path.  TEST INFRASTRUCTURE ONLY:
 * it cross-checks the numpy/scipy oracle with a Faddeeva implementation independent of scipy's (the same algorithm as the
 * device's rarely used general path, so not independent of THAT; the device's hot table path is a different method) and serves as the
 * multi-threaded CPU baseline of bench.py.  Never linked into or called by the product (mc-alf_amd/).
 *
 * Follows /root/reference/mcalf/routines/hires_fitter.py:
 *   voigt_tau            :357-365      tau = cne * Re w(u + i a) / dnu
 *   reconstruct_spec     :409-449      decode, product of exp(-tau) over components x lines + fillers
 *   convolve_model       :452-464      Gaussian LSF, periodic boundary, kernel / sum, top / bot (astropy)
 *   lnlhood_worker       :292-294      -1/2 nansum(ispec2 (obj - model)^2 - ln ispec2 + ln 2 pi)
 * Parity status: pinned through the numpy oracle (tests/test_oracle_c.py: agreement on the reference's
 * fixtures G1/G2 and on random draws).
 *
 * Re w(z): Matta-Reichel / Zaghloul-Ali trapezoid sum with pole correction (step 1/2) for |z| < 8 and the
 * Laplace asymptotic series beyond -- a published algorithm family restated here, not scipy's code.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CCGS 2.9979245e10

typedef struct {
    long npix;
    const double *wl, *flux, *err;   /* obj_wl, obj, obj_noise */
    double velstep;
    int nlines;
    const double *lines;             /* [nlines][3] wrest_A, f, gamma */
    double fill[3];
    int ncompmax, nfill, freespecres, freecont;
    double specres_fixed, contval_fixed;
} oracle_problem;

/* scaled complementary error function exp(y^2) erfc(y), y >= 0 */
static double erfcx_pos(double y) {
    if (y < 25.0) return exp(y * y) * erfc(y);
    double t = 1.0 / (y * y);             /* asymptotic */
    return (1.0 / (y * 1.7724538509055160273)) * (1.0 - 0.5 * t * (1.0 - 1.5 * t * (1.0 - 2.5 * t)));
}

static double re_w_upper(double x, double y);

/* Re w(x + i y) for any real y: lower half plane by w(z) = 2 exp(-z^2) - w(-z) (what scipy's
 * Faddeeva code does; reached when b < 0 makes the damping parameter negative, :360-361). */
static double re_w(double x, double y) {
    x = fabs(x);
    if (!(y < 0.0)) return re_w_upper(x, y);
    const double ay = -y;
    return 2.0 * exp((ay - x) * (x + ay)) * cos(2.0 * x * ay) - re_w_upper(x, ay);
}

static double re_w_upper(double x, double y) {
    const double x2 = x * x, y2 = y * y;
    if (x2 + y2 >= 64.0) {
        const double ir2 = 1.0 / (x2 + y2);
        const double zr = x * ir2, zi = -y * ir2;
        const double sr = zr * zr - zi * zi, si = 2.0 * zr * zi;
        double c[19];
        c[0] = 1.0;
        for (int k = 1; k <= 18; ++k) c[k] = c[k - 1] * (2.0 * k - 1.0) * 0.5;
        double ar = c[18], ai = 0.0;
        for (int k = 17; k >= 0; --k) {
            const double nr = ar * sr - ai * si + c[k];
            const double ni = ar * si + ai * sr;
            ar = nr; ai = ni;
        }
        double H = -0.56418958354775628695 * (zr * ai + zi * ar);
        if (x2 < 745.0 && y < 1e-3) H += exp(y2 - x2) * cos(2.0 * x * y);
        return H;
    }
    const double h = 0.5, G = exp(-x2), xy = x * y, s1 = sin(xy), c2 = cos(2.0 * xy);
    const double T1 = G * erfcx_pos(y) * c2;
    const double T2 = (y > 0.0) ? (2.0 * h) * G * (s1 * s1) / (M_PI * y) : 0.0;
    double S1 = 0.0, S23 = 0.0;
    for (int n = 1; n <= 32; ++n) {
        const double hn = h * n, d = 1.0 / (hn * hn + y2);
        S1 += exp(-hn * hn) * d;
        S23 += (exp(-(hn + x) * (hn + x)) + exp(-(hn - x) * (hn - x))) * d;
    }
    return T1 + T2 + (2.0 * h * y / M_PI) * (-c2 * S1 * G + 0.5 * S23);
}

/* spec[i] *= exp(-tau_i) for one (component, line); hires_fitter.py:357-377 */
static void apply_line(const oracle_problem *pb, double *spec, double logN, double z, double b_kms,
                       const double *line) {
    const double wrest = line[0] / 1e8, f = line[1], gamma = line[2];
    const double cold = pow(10.0, logN), zp1 = z + 1.0, nujk = CCGS / wrest;
    const double dnu = (b_kms * 1e5) / wrest, avoigt = gamma / (4 * M_PI * dnu);
    const double cne = 0.014971475 * cold * f;
    for (long i = 0; i < pb->npix; ++i) {
        const double wave = pb->wl[i] / 1e8;
        const double u = ((CCGS / (wave / zp1)) - nujk) / dnu;
        const double tau = cne * re_w(u, avoigt) / dnu;
        spec[i] *= exp(-1 * tau);
    }
}

/* model spectrum for one parameter vector; `work` holds 2*npix doubles, result in work[0..npix) */
static void reconstruct(const oracle_problem *pb, const double *p, int targonly, double *work) {
    const long n = pb->npix;
    double *spec = work, *out = work + n;
    const int startind = pb->freecont + pb->freespecres, endind = startind + 3 * pb->ncompmax + 1;
    const double R = pb->freespecres ? p[0] : pb->specres_fixed;
    const double cont = pb->freecont ? (pb->freespecres ? p[1] : p[0]) : pb->contval_fixed;
    const int nc = (int)p[startind];
    for (long i = 0; i < n; ++i) spec[i] = 1.0;
    for (int c = 0; c < nc; ++c)
        for (int l = 0; l < pb->nlines; ++l)
            apply_line(pb, spec, p[1 + 3 * c + startind], p[2 + 3 * c + startind], p[3 + 3 * c + startind],
                       pb->lines + 3 * l);
    if (!targonly)
        for (int k = 0; k < pb->nfill; ++k)
            apply_line(pb, spec, p[3 * k + endind], p[3 * k + endind + 1], p[3 * k + endind + 2], pb->fill);
    if (R > pb->velstep) {                       /* :445 */
        const double sigma = (R / 2.354820) / pb->velstep;
        const int half = (int)ceil(3.0348 * sigma), nt = 2 * half + 1;
        double *ker = (double *)malloc(sizeof(double) * nt);
        double ksum = 0.0;
        for (int k = 0; k < nt; ++k) {
            const double xk = (double)(k - half);
            ker[k] = exp(-0.5 * xk * xk / (sigma * sigma)) / (sqrt(2 * M_PI) * sigma);
            ksum += ker[k];
        }
        for (int k = 0; k < nt; ++k) ker[k] /= ksum;
        for (long i = 0; i < n; ++i) {
            double top = 0.0, bot = 0.0;
            for (int k = 0; k < nt; ++k) {
                long j = (i + k - half) % n;
                if (j < 0) j += n;
                top += spec[j] * ker[nt - 1 - k];
                bot += ker[nt - 1 - k];
            }
            out[i] = top / bot;
        }
        free(ker);
        for (long i = 0; i < n; ++i) spec[i] = out[i] * cont;
    } else {
        for (long i = 0; i < n; ++i) spec[i] *= cont;
    }
}

void oracle_model_batch(const oracle_problem *pb, const double *P, long batch, int ndim, int targonly, double *models) {
#pragma omp parallel
    {
        double *work = (double *)malloc(sizeof(double) * 2 * pb->npix);
#pragma omp for schedule(dynamic, 4)
        for (long s = 0; s < batch; ++s) {
            reconstruct(pb, P + s * ndim, targonly, work);
            memcpy(models + s * pb->npix, work, sizeof(double) * pb->npix);
        }
        free(work);
    }
}

void oracle_loglike_batch(const oracle_problem *pb, const double *P, long batch, int ndim, double *logl) {
#pragma omp parallel
    {
        double *work = (double *)malloc(sizeof(double) * 2 * pb->npix);
#pragma omp for schedule(dynamic, 4)
        for (long s = 0; s < batch; ++s) {
            reconstruct(pb, P + s * ndim, 0, work);
            double acc = 0.0;
            for (long i = 0; i < pb->npix; ++i) {
                const double ispec2 = 1.0 / (pb->err[i] * pb->err[i]);
                const double d = pb->flux[i] - work[i];
                const double term = ispec2 * d * d - log(ispec2) + log(2.0 * M_PI);
                if (!isnan(term)) acc += term;      /* np.nansum */
            }
            logl[s] = -0.5 * acc;
        }
        free(work);
    }
}

double oracle_re_w(double x, double y) { return re_w(x, y); }
