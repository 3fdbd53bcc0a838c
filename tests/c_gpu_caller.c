/* The cgo shim minus Go: a plain C99 program that drives the GPU entry points in exactly the order
 * bindings/go/mel/mel_hip.go does -- FromMel (mel_hip.go:175-205: gomel_out_len -> caller-drawn x0 -> gomel_from_mel -> the
 * exp side effect of mel/mel.go:143 on the caller's slice) and FromMelBatch (mel_hip.go:264-317: pointer arrays in C memory,
 * one x0 per utterance, out_len = NULL and out_cap = NULL).  Compiled with gcc and run by tests/test_gpu_round3.py, which
 * writes the inputs and compares the waveform bytes this program writes with tests/golden/golden_v1.npz.
 *
 *   c_gpu_caller NumMels MelFmax Window Resolut Iterations n_pairs n_utt in.bin out.bin
 * in.bin : n_utt x { n_pairs*2 doubles of log-mel, L doubles of x0 }   (x0 stands in for rand.Float64(), mel/mel.go:81-83)
 * out.bin: L doubles from gomel_from_mel on utterance 0, then n_utt x L doubles from gomel_from_mel_batch, then the
 *          n_pairs*2 doubles of utterance 0's slice as the caller finds it afterwards (exp'd). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gomel_hip.h"

#define FAIL(code, what) do { fprintf(stderr, "c_gpu_caller: %s (rc %d: %s)\n", what, (int)(code), gomel_strerror((int)(code))); return 1; } while (0)

int main(int argc, char** argv) {
    gomel_config c;
    gomel_handle* h = NULL;
    size_t n_pairs, n_utt, L, u, i, got = 0;
    double **mels, **x0s, **outs, *single;
    const double **mel_ptrs, **x0_ptrs;
    size_t* lens;
    FILE* f;
    int rc;
    if (argc != 10) { fprintf(stderr, "usage: c_gpu_caller NumMels MelFmax Window Resolut Iterations n_pairs n_utt in.bin out.bin\n"); return 2; }
    if (gomel_default_config(&c) != GOMEL_OK) return 2;                  /* NewMel(), mel/mel.go:30-41 */
    c.num_mels = atoi(argv[1]); c.mel_fmax = atof(argv[2]); c.window = atoi(argv[3]); c.resolut = atoi(argv[4]);
    c.griffin_lim_iterations = atoi(argv[5]);
    n_pairs = (size_t)atol(argv[6]); n_utt = (size_t)atol(argv[7]);
    if (gomel_device_count() < 1) { fprintf(stderr, "c_gpu_caller: no GPU\n"); return 3; }
    L = gomel_out_len(&c, n_pairs);                                       /* mel_hip.go:184 */
    if (L == 0) FAIL(GOMEL_ERR_BAD_LENGTH, "gomel_out_len");
    rc = gomel_create(&c, &h);                                            /* m.acquire() */
    if (rc != GOMEL_OK) FAIL(rc, "gomel_create");

    mels = (double**)malloc(n_utt * sizeof *mels); x0s = (double**)malloc(n_utt * sizeof *x0s); outs = (double**)malloc(n_utt * sizeof *outs);
    single = (double*)malloc(L * sizeof *single);
    f = fopen(argv[8], "rb");
    if (!f || !mels || !x0s || !outs || !single) { fprintf(stderr, "c_gpu_caller: cannot open input / allocate\n"); return 2; }
    for (u = 0; u < n_utt; u++) {
        mels[u] = (double*)malloc(n_pairs * 2 * sizeof(double)); x0s[u] = (double*)malloc(L * sizeof(double)); outs[u] = (double*)calloc(L, sizeof(double));
        if (!mels[u] || !x0s[u] || !outs[u]) return 2;
        if (fread(mels[u], sizeof(double), n_pairs * 2, f) != n_pairs * 2 || fread(x0s[u], sizeof(double), L, f) != L) { fprintf(stderr, "c_gpu_caller: short input\n"); return 2; }
    }
    fclose(f);

    /* (*Mel).FromMel, mel_hip.go:175-205 */
    rc = gomel_from_mel(h, mels[0], n_pairs, x0s[0], 0, single, L, &got);
    if (rc != GOMEL_OK || got != L) FAIL(rc, "gomel_from_mel");

    /* (*Mel).FromMelBatch, mel_hip.go:264-317: the pointer arrays live in C memory */
    mel_ptrs = (const double**)malloc(n_utt * sizeof *mel_ptrs); x0_ptrs = (const double**)malloc(n_utt * sizeof *x0_ptrs);
    lens = (size_t*)malloc(n_utt * sizeof *lens);
    if (!mel_ptrs || !x0_ptrs || !lens) return 2;
    for (u = 0; u < n_utt; u++) { mel_ptrs[u] = mels[u]; x0_ptrs[u] = x0s[u]; lens[u] = n_pairs; }
    rc = gomel_from_mel_batch(h, n_utt, mel_ptrs, lens, x0_ptrs, NULL, outs, NULL);
    if (rc != GOMEL_OK) FAIL(rc, "gomel_from_mel_batch");

    /* Calls in flight: the same call submitted twice as jobs (gomel_from_mel_batch_submit), waited for in reverse order; a handle with a job
     * outstanding refuses gomel_destroy; both jobs deliver the synchronous call's bytes (a call this small runs inside submit). */
    {
        gomel_job *j1 = NULL, *j2 = NULL;
        double **o1 = (double**)malloc(n_utt * sizeof *o1), **o2 = (double**)malloc(n_utt * sizeof *o2);
        int fin = 0;
        if (!o1 || !o2) return 2;
        for (u = 0; u < n_utt; u++) { o1[u] = (double*)calloc(L, sizeof(double)); o2[u] = (double*)calloc(L, sizeof(double)); if (!o1[u] || !o2[u]) return 2; }
        rc = gomel_from_mel_batch_submit(h, n_utt, mel_ptrs, lens, x0_ptrs, NULL, o1, NULL, &j1);
        if (rc != GOMEL_OK || !j1) FAIL(rc, "gomel_from_mel_batch_submit (1)");
        rc = gomel_from_mel_batch_submit(h, n_utt, mel_ptrs, lens, x0_ptrs, NULL, o2, NULL, &j2);
        if (rc != GOMEL_OK || !j2) FAIL(rc, "gomel_from_mel_batch_submit (2)");
        if (gomel_destroy(h) != GOMEL_ERR_BUSY) FAIL(GOMEL_ERR_BUSY, "gomel_destroy with jobs outstanding must be refused with GOMEL_ERR_BUSY");
        rc = gomel_job_wait(j2);
        if (rc != GOMEL_OK) FAIL(rc, "gomel_job_wait (2)");
        rc = gomel_job_poll(j1, &fin);
        if (rc != GOMEL_OK) FAIL(rc, "gomel_job_poll");
        rc = gomel_job_wait(j1);
        if (rc != GOMEL_OK) FAIL(rc, "gomel_job_wait (1)");
        for (u = 0; u < n_utt; u++)
            if (memcmp(o1[u], outs[u], L * sizeof(double)) != 0 || memcmp(o2[u], outs[u], L * sizeof(double)) != 0) FAIL(GOMEL_ERR_BAD_CONFIG, "a job's waveform differs from the synchronous call's");
    }

    /* the side effect of spectral_denormalize on the caller's slice (mel/mel.go:143, mel/impl.go:421-427): done by the shim, after the call */
    for (i = 0; i < n_pairs * 2; i++) mels[0][i] = exp(mels[0][i]);

    f = fopen(argv[9], "wb");
    if (!f) return 2;
    fwrite(single, sizeof(double), L, f);
    for (u = 0; u < n_utt; u++) fwrite(outs[u], sizeof(double), L, f);
    fwrite(mels[0], sizeof(double), n_pairs * 2, f);
    fclose(f);
    rc = gomel_destroy(h);
    if (rc != GOMEL_OK) FAIL(rc, "gomel_destroy");
    printf("c gpu caller ok: %lu utterances x %lu samples\n", (unsigned long)n_utt, (unsigned long)L);
    return 0;
}
