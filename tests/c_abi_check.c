/* Compiled as C99 by tests/test_abi.py: the header must be plain C (cgo compiles it as C), and the host-only entry points
 * must work from a C caller without a GPU. */
#include <stdio.h>
#include <string.h>

#include "gomel_hip.h"

int main(void) {
    gomel_config c;
    if (gomel_default_config(&c) != GOMEL_OK) return 1;
    if (c.num_mels != 160 || c.window != 256 || c.resolut != 2048 || c.griffin_lim_iterations != 2) return 2; /* mel/mel.go:30-41 */
    c.num_mels = 192; c.window = 1280; c.resolut = 4096;
    if (gomel_out_len(&c, (size_t)342 * 192) != 440576) return 3;
    if (gomel_pad_len(441000, 1280) != 441599 || !gomel_is_padded(441000, 441599, 1280)) return 4;       /* mel/impl.go:429-479 */
    if (gomel_to_mel_pairs(&c, 441000) != (size_t)342 * 192) return 5;
    if (gomel_phase_pairs(&c, 768, 441000) != (size_t)342 * 768 || gomel_phase_out_len(&c, 768, (size_t)342 * 768) != 440576) return 6;
    if (strcmp(gomel_strerror(GOMEL_OK), "ok") != 0 || gomel_abi_version() != GOMEL_HIP_ABI_VERSION) return 7;
    /* ABI v2 entry points as a C caller sees them: the generic-length gate, the table builder, the multi-handle batch call's
     * argument checks -- all of it without a GPU */
    c.resolut = 3000; if (gomel_tables_bytes_for(&c) == 0) return 10;            /* any length up to 4096 */
    c.resolut = 3001; if (gomel_tables_bytes_for(&c) == 0) return 11;            /* odd lengths too */
    c.resolut = 6000; if (gomel_tables_bytes_for(&c) == 0) return 17;            /* ... and up to 8192 (Bluestein in two halves) */
    c.resolut = 70000; if (gomel_tables_bytes_for(&c) != 0) return 22;           /* above 65536: refused */
    c.resolut = 8192; if (gomel_tables_bytes_for(&c) == 0) return 12;
    c.resolut = 4096;
    {
        const double* mel[1] = {NULL}; double* out[1] = {NULL}; size_t np[1] = {192};
        if (gomel_from_mel_batch_multi(NULL, 1, 1, mel, np, NULL, NULL, out, NULL) != GOMEL_ERR_NULL) return 13;
        if (gomel_istft_spec(NULL, NULL, 1, 1, NULL, 0, NULL, 0, NULL, NULL) != GOMEL_ERR_NULL) return 14;
        if (gomel_to_mel_batch(NULL, 1, mel, np, out, NULL, NULL) != GOMEL_ERR_NULL) return 15;
        if (gomel_host_free(NULL) != GOMEL_OK || gomel_host_register(NULL, 16) != GOMEL_ERR_NULL) return 16;
    }
    {   /* the pipeline's chunk planner is host logic: 64 utterances of 400 frames at the ToWavPng geometry -> half + full + half */
        size_t np[64], bounds[66], nch = 0, u;
        for (u = 0; u < 64; u++) np[u] = (size_t)400 * 192;
        if (gomel_plan_chunks(&c, 64, np, bounds, 66, &nch) != GOMEL_OK) return 18;
        if (nch < 2 || bounds[0] != 0 || bounds[nch] != 64) return 19;
        for (u = 0; u < nch; u++) if (bounds[u + 1] <= bounds[u]) return 20;
        if (gomel_plan_chunks(&c, 64, np, bounds, 1, &nch) != GOMEL_ERR_BUFFER_TOO_SMALL) return 21;
    }
    gomel_handle* h = NULL;
    int rc = gomel_create(&c, &h);
    if (gomel_device_count() == 0) {
        if (rc != GOMEL_ERR_NO_DEVICE || h != NULL) return 8;   /* no GPU: refuse, never fall back */
    } else {
        if (rc != GOMEL_OK || h == NULL) return 9;
        gomel_destroy(h);
    }
    printf("c abi ok (devices: %d)\n", gomel_device_count());
    return 0;
}
