// C caller of gomel_from_mel_batch (what a cgo host does): U utterances of 400 frames, pinned (mode 0) or pageable (1) buffers,
// geometry 0 = ToWavPng, 1 = TTS, 2 = NewMel defaults, 3 = 8 kHz.  Built and used by scripts/chunk_plans.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <chrono>
#include "gomel_hip.h"
int main(int argc, char** argv) {
    int U = argc > 1 ? atoi(argv[1]) : 64, mode = argc > 2 ? atoi(argv[2]) : 0, geo = argc > 3 ? atoi(argv[3]) : 0;
    gomel_config c; gomel_default_config(&c);
    c.num_mels = 192; c.mel_fmax = 16000; c.window = 1280; c.resolut = 4096; c.griffin_lim_iterations = 60;
    if (geo == 1) { c.num_mels = 80; c.mel_fmax = 8000; c.window = 256; c.resolut = 1024; }
    if (geo == 2) { c.num_mels = 160; c.mel_fmax = 8000; c.window = 256; c.resolut = 2048; }
    if (geo == 3) { c.num_mels = 40; c.mel_fmax = 4000; c.window = 80; c.resolut = 256; }
    gomel_handle* h = nullptr;
    if (gomel_create(&c, &h)) { printf("create failed\n"); return 1; }
    size_t F = 400, np_ = F * c.num_mels, L = gomel_out_len(&c, np_);
    double *mel, *out;
    if (mode == 0) { mel = (double*)gomel_host_alloc(U * np_ * 16); out = (double*)gomel_host_alloc(U * L * 8); }
    else { mel = (double*)malloc(U * np_ * 16); out = (double*)malloc(U * L * 8); memset(out, 0, U * L * 8); }
    for (size_t i = 0; i < U * np_ * 2; i++) mel[i] = -11.5 + 15.4 * ((i * 2654435761u % 1000003) / 1000003.0);
    std::vector<const double*> mp(U); std::vector<double*> op(U); std::vector<size_t> n(U, np_), cap(U, L); std::vector<uint64_t> seeds(U);
    for (int u = 0; u < U; u++) { mp[u] = mel + u * np_ * 2; op[u] = out + u * L; seeds[u] = u; }
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        int rc = gomel_from_mel_batch(h, U, mp.data(), n.data(), nullptr, seeds.data(), op.data(), cap.data());
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("rc %d: %.2f ms, %.3f M frames/s  out[5]=%g\n", rc, dt * 1e3, U * F / dt / 1e6, out[5]);
    }
    gomel_destroy(h);
    return 0;
}
