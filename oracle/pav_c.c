/* oracle/pav_c.c - C restatement of the exact stack PAV of oracle/pav.py
 * (pav_exact_py), used so the CPU oracle finishes in seconds at n ~ 1e6.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): never linked into, loaded
 * by or shipped with the product library.
 *
 * Problem (reference: src/optim/algorithms.py:92-101, src/util/pav.py:93-178):
 *   min_{u_1<=...<=u_n} sum_i sigma_i*loss(u_i) + rho/2 (u_i - m_i)^2, m ascending.
 * Block value = root of mean(sigma)*loss'(x) + rho*(x - mean(m)) = 0
 * (src/util/pav.py:134-140; element solve src/util/individual_solver.py:60-80).
 * loss 0 = binary cross entropy (loss' = sigmoid), 1 = hinge (closed form, the
 * limit of the bisection in src/util/individual_solver.py:11-42).
 */
#include <math.h>
#include <stdlib.h>

static double sig(double x) {
    if (x > 0) return 1.0 / (1.0 + exp(-x));
    double e = exp(x);
    return e / (1.0 + e);
}

static double block_value(int loss, double ssig, double sm, double cnt, double rho) {
    double mbar = sm / cnt, sbar = ssig / cnt;
    if (loss == 1) {
        double a = mbar - sbar / rho;
        if (a >= -1.0) return a;
        return mbar <= -1.0 ? mbar : -1.0;
    }
    double lo = mbar - sbar / rho, hi = mbar, x = hi;
    double dxold = hi - lo, dx = dxold;
    double s = sig(x);
    double g = sbar * s + rho * (x - mbar);
    double h = sbar * s * (1.0 - s) + rho;
    for (int it = 0; it < 200; ++it) {
        double xn;
        if (g == 0) break;
        if (((x - hi) * h - g) * ((x - lo) * h - g) > 0 || fabs(2.0 * g) > fabs(dxold * h)) {
            dxold = dx; dx = 0.5 * (hi - lo); xn = lo + dx;
        } else {
            dxold = dx; dx = g / h; xn = x - dx;
        }
        if (xn == x) break;
        x = xn;
        s = sig(x);
        g = sbar * s + rho * (x - mbar);
        h = sbar * s * (1.0 - s) + rho;
        if (g < 0) lo = x; else hi = x;
    }
    return x;
}

long oracle_pav_exact(int loss, const double *sigma, const double *m, long n,
                      double rho, double *out) {
    if (n <= 0) return 0;
    double *S = malloc(sizeof(double) * n), *M = malloc(sizeof(double) * n),
           *X = malloc(sizeof(double) * n);
    long *C = malloc(sizeof(long) * n);
    if (!S || !M || !X || !C) { free(S); free(M); free(X); free(C); return -1; }
    long top = 0;
    for (long i = 0; i < n; ++i) {
        double s = sigma[i], mm = m[i], x = block_value(loss, s, mm, 1.0, rho);
        long c = 1;
        while (top > 0 && X[top - 1] > x) {
            --top;
            s += S[top]; mm += M[top]; c += C[top];
            x = block_value(loss, s, mm, (double)c, rho);
        }
        S[top] = s; M[top] = mm; C[top] = c; X[top] = x; ++top;
    }
    long k = 0;
    for (long b = 0; b < top; ++b)
        for (long j = 0; j < C[b]; ++j) out[k++] = X[b];
    free(S); free(M); free(X); free(C);
    return top;
}
