// C caller of gomel_to_mel_batch: U utterances of 10 s at 44.1 kHz (441 000 samples), ToWavPng geometry, pinned (mode 0) or
// pageable (1) buffers.  usage: lib_tomel [U] [mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <chrono>
#include "gomel_hip.h"
int main(int argc, char** argv) {
    int U = argc > 1 ? atoi(argv[1]) : 64, mode = argc > 2 ? atoi(argv[2]) : 0;
    gomel_config c; gomel_default_config(&c);
    c.num_mels = 192; c.mel_fmax = 16000; c.window = 1280; c.resolut = 4096;
    gomel_handle* h = nullptr;
    if (gomel_create(&c, &h)) { printf("create failed\n"); return 1; }
    const size_t n = 441000, pairs = gomel_to_mel_pairs(&c, n);
    double *wav, *out;
    if (mode == 0) { wav = (double*)gomel_host_alloc(U * n * 8); out = (double*)gomel_host_alloc(U * pairs * 16); }
    else { wav = (double*)malloc(U * n * 8); out = (double*)malloc(U * pairs * 16); memset(out, 0, U * pairs * 16); }
    for (size_t i = 0; i < U * n; i++) wav[i] = 0.3 * sin(0.01 * (double)(i % n) + 1e-7 * (double)(i % n) * (double)(i % n)) + 1e-3 * ((i * 2654435761u % 1000) / 1000.0);
    std::vector<const double*> wp(U); std::vector<double*> op(U); std::vector<size_t> ns(U, n), cap(U, pairs), got(U);
    for (int u = 0; u < U; u++) { wp[u] = wav + u * n; op[u] = out + u * pairs * 2; }
    for (int rep = 0; rep < 4; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        int rc = gomel_to_mel_batch(h, U, wp.data(), ns.data(), op.data(), cap.data(), got.data());
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("rc %d: %.2f ms, %.3f M frames/s (%.0f x real time), out[7]=%g\n", rc, dt * 1e3, U * (pairs / 192) / dt / 1e6, U * 10.0 / dt, out[7]);
    }
    gomel_destroy(h);
    return 0;
}
