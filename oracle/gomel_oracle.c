/* CPU oracle, C restatement of gomel's FromMel / ToMel path.  TEST INFRASTRUCTURE ONLY.
 *
 * Second, independent restatement of the reference algorithm (the first is gomel_oracle.py).  It keeps
 * the reference's COST STRUCTURE so that it can serve as the timed CPU baseline ("port") in bench.py:
 * per frame a full N-point complex128 radix-2 FFT of the windowed frame, per-bin hypot / atan2 / sincos
 * (Go cmplx.Abs / cmplx.Phase / cmplx.Rect), Hermitian fill, a full N-point inverse FFT scaled 1/N,
 * synthesis window and overlap-add without window-sum normalisation.  Nothing under gomel_amd/ links,
 * loads or calls this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED: the reference cannot be compiled or run here (Go; FFT/STFT in un-vendored modules
 * go-dsp v0.0.0-20180508042940-11479a337f12 and gossp v0.0.1) and has no tests or golden vectors for this
 * path.  This file is pinned only against gomel_oracle.py (agreement ~1e-10), numpy.fft, and tests/golden/.
 *
 * file:line citations are relative to /root/reference.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    /* mel/mel.go:10-27 */
    int32_t num_mels;
    int32_t window;   /* hop (gossp frameShift) */
    int32_t resolut;  /* frame / FFT length */
    int32_t iterations;
    double mel_fmin, mel_fmax, tune_mul, tune_add;
} oracle_cfg;

enum { ORACLE_OK = 0, ORACLE_BAD_LENGTH = 1, ORACLE_BAD_N = 2, ORACLE_OOM = 3, ORACLE_INDEX = 4 };

/* ---- radix-2 complex128 FFT, X[k] = sum x[n] e^{-+2 pi i nk/N} (go-dsp fft.FFT for power-of-two N) ---- */
static void fft_radix2(double* re, double* im, int n, const double* twr, const double* twi, int inverse) {
    for (int i = 1, j = 0; i < n; i++) { /* bit reversal */
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int i = 0; i < n; i += len) {
            for (int k = 0; k < half; k++) {
                double wr = twr[k * step], wi = inverse ? -twi[k * step] : twi[k * step];
                double ur = re[i + k], ui = im[i + k];
                double vr = re[i + k + half] * wr - im[i + k + half] * wi;
                double vi = re[i + k + half] * wi + im[i + k + half] * wr;
                re[i + k] = ur + vr; im[i + k] = ui + vi;
                re[i + k + half] = ur - vr; im[i + k + half] = ui - vi;
            }
        }
    }
}

static int is_pow2(int n) { return n >= 2 && (n & (n - 1)) == 0; }

/* exported for the DFT-definition test */
int oracle_fft(double* re, double* im, int n, int inverse) {
    if (!is_pow2(n)) return ORACLE_BAD_N;
    double* tw = (double*)malloc(sizeof(double) * (size_t)n);
    if (!tw) return ORACLE_OOM;
    for (int k = 0; k < n / 2; k++) {
        tw[k] = cos(-2.0 * M_PI * k / n);
        tw[n / 2 + k] = sin(-2.0 * M_PI * k / n);
    }
    fft_radix2(re, im, n, tw, tw + n / 2, inverse);
    if (inverse) for (int i = 0; i < n; i++) { re[i] /= n; im[i] /= n; }
    free(tw);
    return ORACLE_OK;
}

void oracle_hann(double* w, int n, int symmetric) {
    /* gossp window.CreateHanning (believed symmetric) */
    double arg = 2.0 * M_PI / (double)(symmetric ? n - 1 : n);
    for (int i = 0; i < n; i++) w[i] = 0.5 - 0.5 * cos(arg * (double)i);
}

static double hz_to_mel(double v) { return 1127.0 * log(1.0 + (v / 700.0)); } /* impl.go:304-308 */
static double mel_to_hz(double v) { return 700.0 * (exp(v / 1127.0) - 1.0); } /* impl.go:298-302 */

size_t oracle_out_len(const oracle_cfg* c, size_t n_pairs) { /* mel.go:80 */
    size_t F = n_pairs / (size_t)c->num_mels;
    return (size_t)c->resolut + (F - 1) * (size_t)c->window;
}

/* undomel, mel/impl.go:347-384, literal: the index math is recomputed for every frame as the reference does. */
static int undomel(int filtersize, int mels, const double* mel, size_t n_pairs, double fmin, double fmax, double* out) {
    double filterbin = hz_to_mel(fmax) / (double)mels;
    size_t o = 0;
    for (size_t j = 0; j < n_pairs; j += (size_t)mels) {
        for (int i = 0; i < filtersize; i++) {
            double vallo = hz_to_mel(((double)i * (fmax + fmin) / (double)filtersize) - fmin) / filterbin;
            double valhi = hz_to_mel(((double)(i + 1) * (fmax + fmin) / (double)filtersize) - fmin) / filterbin;
            double inlo, modlo = modf(vallo, &inlo);
            double inhi = floor(valhi);
            if (inlo < 0) { inlo = 0; modlo = 0; inhi = 0; }
            long lo = (long)inlo, hi = (long)inhi;
            for (int l = 0; l < 2; l++) {
                double total = 0;
                if (lo == hi) {
                    if (lo < 0 || lo >= mels) return ORACLE_INDEX;
                    total += mel[(j + (size_t)lo) * 2 + l];
                } else if (lo + 1 == hi && hi < mels) {
                    total += mel[(j + (size_t)lo) * 2 + l] * (1 - modlo);
                    total += mel[(j + (size_t)hi) * 2 + l] * modlo;
                } else {
                    for (long k = lo; k < hi; k++) {
                        if (k < 0 || k >= mels) return ORACLE_INDEX;
                        total += mel[(j + (size_t)k) * 2 + l];
                    }
                    total /= inhi - inlo + 1;
                }
                out[o * 2 + l] = total;
            }
            o++;
        }
    }
    return ORACLE_OK;
}

/* domel, mel/impl.go:310-345 */
static int domel(int filtersize, int mels, const double* spec, size_t n, double fmin, double fmax, double* out) {
    double melbin = hz_to_mel(fmax) / (double)mels;
    size_t o = 0;
    for (size_t j = 0; j < n; j += (size_t)filtersize) {
        for (int i = 0; i < mels; i++) {
            double vallo = (double)filtersize * (fmin + mel_to_hz(melbin * (double)i)) / (fmax + fmin);
            double valhi = (double)filtersize * (fmin + mel_to_hz(melbin * (double)(i + 1))) / (fmax + fmin);
            double inlo, modlo = modf(vallo, &inlo);
            double inhi = floor(valhi);
            if (inlo < 0) { inlo = 0; modlo = 0; inhi = 0; }
            long lo = (long)inlo, hi = (long)inhi;
            for (int l = 0; l < 2; l++) {
                double total = 0;
                if (lo + 1 == hi) {
                    if (hi >= filtersize) return ORACLE_INDEX;
                    total += spec[(j + (size_t)lo) * 2 + l] * (1 - modlo);
                    total += spec[(j + (size_t)hi) * 2 + l] * modlo;
                } else {
                    if (hi > filtersize) return ORACLE_INDEX;
                    for (long k = lo; k < hi; k++) total += spec[(j + (size_t)k) * 2 + l];
                    total /= (double)(hi - lo + 1);
                }
                out[o * 2 + l] = total;
            }
            o++;
        }
    }
    return ORACLE_OK;
}

/* (*Mel).FromMel, mel/mel.go:142-152, with ISTFT (mel.go:76-139) inlined.
 * mel: [n_pairs][2] log-mel, exp'd IN PLACE (impl.go:421-427).  window: N doubles.  x0: L doubles, replaces
 * rand.Float64() (mel.go:81-83).  out: L doubles. */
int oracle_from_mel(const oracle_cfg* c, double* mel, size_t n_pairs, const double* window, const double* x0, double* out) {
    const int N = c->resolut, hop = c->window, K = c->num_mels, H = N / 2;
    if (!is_pow2(N)) return ORACLE_BAD_N;
    if (n_pairs == 0 || n_pairs % (size_t)K != 0) return ORACLE_BAD_LENGTH;
    const size_t F = n_pairs / (size_t)K, L = (size_t)N + (F - 1) * (size_t)hop;
    for (size_t i = 0; i < n_pairs * 2; i++) mel[i] = exp(mel[i]);            /* mel.go:143 */

    double* lin = (double*)malloc(sizeof(double) * F * (size_t)H * 2);
    double* sre = (double*)malloc(sizeof(double) * F * (size_t)N);
    double* sim = (double*)malloc(sizeof(double) * F * (size_t)N);
    double* x = (double*)malloc(sizeof(double) * L);
    double* xn = (double*)malloc(sizeof(double) * L);
    double* fr = (double*)malloc(sizeof(double) * (size_t)N * 3);
    if (!lin || !sre || !sim || !x || !xn || !fr) { free(lin); free(sre); free(sim); free(x); free(xn); free(fr); return ORACLE_OOM; }
    double *fi = fr + N, *tw = fr + 2 * N;
    for (int k = 0; k < H; k++) { tw[k] = cos(-2.0 * M_PI * k / N); tw[H + k] = sin(-2.0 * M_PI * k / N); }

    int rc = undomel(H, K, mel, n_pairs, c->mel_fmin, c->mel_fmax, lin);     /* mel.go:147 */
    if (rc == ORACLE_OK) {
        for (size_t f = 0; f < F; f++)                                          /* undospectrum, impl.go:386-408 */
            for (int j = 0; j < H; j++) {
                double r0 = (lin[(f * H + j) * 2 + 0] - c->tune_add) / c->tune_mul;
                double r1 = (lin[(f * H + j) * 2 + 1] - c->tune_add) / c->tune_mul;
                sre[f * N + j] = r0; sim[f * N + j] = 0;
                sre[f * N + (N - j - 1)] = r1; sim[f * N + (N - j - 1)] = 0;
            }
        memcpy(x, x0, sizeof(double) * L);                                      /* mel.go:80-83 */
        for (int it = 0; it < c->iterations; it++) {                            /* mel.go:85 */
            for (size_t f = 0; f < F; f++) {                                    /* mel.go:87-109 */
                for (int j = 0; j < N; j++) { fr[j] = x[f * hop + j] * window[j]; fi[j] = 0; }
                fft_radix2(fr, fi, N, tw, tw + H, 0);                           /* fft.FFTReal */
                double *pr = sre + f * N, *pi = sim + f * N;
                for (int j = 0; j < N; j++) {
                    double mag = hypot(pr[j], pi[j]);                           /* cmplx.Abs */
                    double ph = atan2(fi[j], fr[j]);                            /* cmplx.Phase */
                    double s, co; sincos(ph, &s, &co);                          /* cmplx.Rect */
                    pr[j] = mag * co; pi[j] = mag * s;
                }
                for (int j = 1; j < H; j++) { pr[N - j] = pr[j]; pi[N - j] = -pi[j]; } /* mel.go:105-108 */
            }
            memset(xn, 0, sizeof(double) * L);                                  /* mel.go:112 */
            for (size_t f = 0; f < F; f++) {                                    /* mel.go:115-125 */
                memcpy(fr, sre + f * N, sizeof(double) * N);
                memcpy(fi, sim + f * N, sizeof(double) * N);
                fft_radix2(fr, fi, N, tw, tw + H, 1);                           /* fft.IFFT */
                for (int j = 0; j < N; j++) xn[f * hop + j] += (fr[j] / N) * window[j];
            }
            double* t = x; x = xn; xn = t;                                      /* mel.go:135 */
        }
        memcpy(out, x, sizeof(double) * L);
    }
    free(lin); free(sre); free(sim); free(x); free(xn); free(fr);
    return rc;
}

/* Utterance-parallel batch: the reference is serial; this only spreads independent FromMel calls over
 * `threads` host threads for the "all host cores" baseline line.  Returns the number of threads used (<0: error). */
int oracle_from_mel_batch(const oracle_cfg* c, int n_utt, double* const* mel, const size_t* n_pairs,
                          const double* window, const double* const* x0, double* const* out, int threads) {
    int rc = 0, used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
        for (int u = 0; u < n_utt; u++) {
            int r = oracle_from_mel(c, mel[u], n_pairs[u], window, x0[u], out[u]);
            if (r != ORACLE_OK) {
#pragma omp critical
                rc = r;
            }
        }
    }
#else
    (void)threads;
    for (int u = 0; u < n_utt; u++) { int r = oracle_from_mel(c, mel[u], n_pairs[u], window, x0[u], out[u]); if (r) rc = r; }
#endif
    return rc ? -rc : used;
}

/* pad, mel/impl.go:429-455: returns the padded length */
size_t oracle_pad_len(size_t n, int filt) {
    size_t mint = 15 * (size_t)filt, padlen = 0;
    if (n >= mint) {
        size_t rem = (n - mint) % (size_t)filt;
        if (rem != 0) padlen = (size_t)filt - rem - 1;
    } else {
        padlen = mint - n - 1;
    }
    return n + padlen;
}

/* number of [2]float64 pairs ToMel returns for n input samples */
size_t oracle_to_mel_pairs(const oracle_cfg* c, size_t n) {
    size_t p = oracle_pad_len(n, c->window);
    if (p < (size_t)c->resolut) return 0;
    long F = (long)((double)(p - (size_t)c->resolut) / (double)c->window) + 1; /* gossp NumFrames */
    return (size_t)F * (size_t)c->num_mels;
}

/* (*Mel).ToMel, mel/mel.go:46-74.  out: oracle_to_mel_pairs(n) pairs. */
int oracle_to_mel(const oracle_cfg* c, const double* buf, size_t n, const double* window, double* out) {
    const int N = c->resolut, hop = c->window, K = c->num_mels, H = N / 2;
    if (!is_pow2(N)) return ORACLE_BAD_N;
    size_t P = oracle_pad_len(n, hop);
    if (P < (size_t)N) return ORACLE_BAD_LENGTH;
    size_t F = (size_t)((long)((double)(P - (size_t)N) / (double)hop) + 1);
    double* pb = (double*)calloc(P, sizeof(double));
    double* sp = (double*)malloc(sizeof(double) * F * (size_t)H * 2);
    double* fr = (double*)malloc(sizeof(double) * (size_t)N * 3);
    if (!pb || !sp || !fr) { free(pb); free(sp); free(fr); return ORACLE_OOM; }
    memcpy(pb, buf, sizeof(double) * n);                                        /* mel.go:48 */
    double *fi = fr + N, *tw = fr + 2 * N;
    for (int k = 0; k < H; k++) { tw[k] = cos(-2.0 * M_PI * k / N); tw[H + k] = sin(-2.0 * M_PI * k / N); }
    for (size_t f = 0; f < F; f++) {                                            /* gossp STFT, mel.go:52 */
        for (int j = 0; j < N; j++) { fr[j] = pb[f * hop + j] * window[j]; fi[j] = 0; }
        fft_radix2(fr, fi, N, tw, tw + H, 0);
        for (int j = 0; j < H; j++) {                                           /* mel.go:55-63 */
            sp[(f * H + j) * 2 + 0] = hypot(fr[j], fi[j]);
            sp[(f * H + j) * 2 + 1] = hypot(fr[N - j - 1], fi[N - j - 1]);
        }
    }
    int rc = domel(H, K, sp, F * (size_t)H, c->mel_fmin, c->mel_fmax, out);     /* mel.go:68 */
    if (rc == ORACLE_OK)
        for (size_t i = 0; i < F * (size_t)K * 2; i++) {                        /* impl.go:410-419 */
            double v = out[i];
            if (v < 1e-5) v = 1e-5;
            out[i] = log(v);
        }
    free(pb); free(sp); free(fr);
    return rc;
}
