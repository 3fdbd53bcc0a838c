// gomel_host.cpp -- the C ABI's HOST-ONLY entry points (include/gomel_hip.h: no device is touched by any of them) and the diagnostics
// of the host logic (fake pin backend, pool self-test) that tests/host_stress.cpp and the sanitizer builds drive.  Plain C++17: built
// with g++ into libgomel_hip.so beside gomel_hip.hip, and alone (libgomel_host.so) by scripts/sanitize_host.sh.
#include "gomel_host.hpp"

#include <pthread.h>
#include <sched.h>

extern "C" {

int gomel_abi_version(void) { return GOMEL_HIP_ABI_VERSION; }

const char* gomel_strerror(int s) {
    switch (s) {
        case GOMEL_OK: return "ok";
        case GOMEL_ERR_BAD_CONFIG: return "bad configuration (NumMels/Window/Resolut/TuneMul/MelFmax)";
        case GOMEL_ERR_UNSUPPORTED_N: return "Resolut not covered: any length from 4 to 65536 is (FFT test hooks, target magnitudes and package phase: 256 ... 4096, powers of two)";
        case GOMEL_ERR_BAD_LENGTH: return "spectrogram length is not a multiple of NumMels";
        case GOMEL_ERR_EMPTY: return "empty input";
        case GOMEL_ERR_TABLE_RANGE: return "mel table indexes outside the band/bin range for this configuration";
        case GOMEL_ERR_BUFFER_TOO_SMALL: return "output buffer too small";
        case GOMEL_ERR_NULL: return "null pointer or call out of order";
        case GOMEL_ERR_HIP: return "HIP runtime error (see gomel_last_hip_error)";
        case GOMEL_ERR_NO_DEVICE: return "no usable HIP device";
        case GOMEL_ERR_BAD_TABLES: return "table blob does not match the configuration";
        case GOMEL_ERR_TOO_SHORT: return "padded input shorter than one frame";
        case GOMEL_ERR_NO_MEMORY: return "host allocation failed";
        case GOMEL_ERR_BUSY: return "handle busy: submitted jobs have not been waited for";
        default: return "unknown status";
    }
}

int gomel_default_config(gomel_config* c) {
    if (!c) return GOMEL_ERR_NULL;
    c->num_mels = 160; c->window = 256; c->resolut = 2048; c->griffin_lim_iterations = 2;  // mel/mel.go:30-41
    c->mel_fmin = 0; c->mel_fmax = 8000; c->tune_mul = 1; c->tune_add = 0;
    c->device = -1; c->hann_periodic = 0;
    return GOMEL_OK;
}

size_t gomel_tables_bytes_for(const gomel_config* c) {
    if (validate_config(c) != GOMEL_OK) return 0;
    return sizeof(TableHeader) + (size_t)c->resolut * 8 + ((size_t)c->resolut / 2 + c->num_mels) * sizeof(BinRow);
}

int gomel_tables_build(const gomel_config* cfg, void* blob, size_t cap) {
    if (!cfg || !blob) return GOMEL_ERR_NULL;
    int rc = validate_config(cfg);
    if (rc) return rc;
    if (cap < gomel_tables_bytes_for(cfg)) return GOMEL_ERR_BUFFER_TOO_SMALL;
    GOMEL_TRY
    std::vector<double> window;
    std::vector<BinRow> undo, dom;
    int dom_status = GOMEL_OK;
    rc = build_host_tables(*cfg, window, undo, dom, &dom_status);
    if (rc) return rc;
    pack_tables(*cfg, dom_status, window, undo, dom, blob);
    return GOMEL_OK;
    GOMEL_CATCH
}

int32_t gomel_auto_fp64_iterations(const gomel_config* cfg, int32_t iterations) {
    if (!cfg || iterations <= 0) return 0;
    if (!gomel_supported_n(cfg->resolut)) return iterations;   // generic-length kernels: fp64
    return auto_fp64_iterations(*cfg, iterations);
}
size_t gomel_out_len(const gomel_config* c, size_t n_pairs) {
    if (!c || c->num_mels <= 0 || n_pairs == 0 || n_pairs % (size_t)c->num_mels) return 0;
    return (size_t)c->resolut + (n_pairs / (size_t)c->num_mels - 1) * (size_t)c->window;  // mel/mel.go:80
}

size_t gomel_pad_len(size_t n, int32_t window) {  // mel/impl.go:429-455
    if (window <= 0) return n;
    size_t mint = 15 * (size_t)window, padlen = 0;
    if (n >= mint) {
        size_t rem = (n - mint) % (size_t)window;
        if (rem != 0) padlen = (size_t)window - rem - 1;
    } else {
        padlen = mint - n - 1;
    }
    return n + padlen;
}

int gomel_is_padded(size_t orig, size_t padded, int32_t window) {  // mel/impl.go:457-479
    return gomel_pad_len(orig, window) == padded ? 1 : 0;
}

size_t gomel_istft_row_len(const gomel_config* cfg) { return (cfg && cfg->resolut > 0) ? istft_row(cfg->resolut) : 0; }
size_t gomel_to_mel_pairs(const gomel_config* c, size_t n) {
    if (!c || c->window <= 0 || c->resolut <= 0 || c->num_mels <= 0) return 0;
    size_t p = gomel_pad_len(n, c->window);
    if (p < (size_t)c->resolut) return 0;
    long F = (long)((double)(p - (size_t)c->resolut) / (double)c->window) + 1;  // gossp NumFrames
    return (size_t)F * (size_t)c->num_mels;
}

// (*Mel).ToMel for many waveforms in one go: every utterance padded as mel/impl.go:429-455 does, all frames through one STFT
// launch and one domel launch, log-mel pairs back per utterance.  The single-utterance entry point is the n_utt = 1 case.
size_t gomel_phase_pairs(const gomel_config* c, int32_t num_freqs, size_t n) {   // len(ToPhase(buf))
    if (!c || c->window <= 0 || c->resolut <= 0 || num_freqs <= 0) return 0;
    size_t p = gomel_pad_len(n, c->window);
    if (p < (size_t)c->resolut) return 0;
    long F = (long)((double)(p - (size_t)c->resolut) / (double)c->window) + 1;
    size_t nf = (size_t)std::min<int64_t>(num_freqs, c->resolut / 2);
    return (size_t)F * nf;
}

size_t gomel_phase_out_len(const gomel_config* c, int32_t num_freqs, size_t n_pairs) {
    if (!c || num_freqs <= 0 || n_pairs == 0 || n_pairs % (size_t)num_freqs) return 0;
    return (size_t)c->resolut + (n_pairs / (size_t)num_freqs - 1) * (size_t)c->window;
}

int gomel_plan_chunks_mode(const gomel_config* cfg, int32_t fp64_iterations, size_t n_utt, const size_t* n_pairs, size_t* bounds, size_t cap, size_t* n_chunks) {
    if (!cfg || !n_pairs || !bounds || !n_chunks) return GOMEL_ERR_NULL;
    if (n_utt == 0) return GOMEL_ERR_EMPTY;
    if (cfg->num_mels <= 0 || cfg->window <= 0 || cfg->resolut < 4) return GOMEL_ERR_BAD_CONFIG;
    for (size_t u = 0; u < n_utt; u++) {
        if (n_pairs[u] == 0) return GOMEL_ERR_EMPTY;
        if (n_pairs[u] % (size_t)cfg->num_mels) return GOMEL_ERR_BAD_LENGTH;
    }
    GOMEL_TRY
    if (fp64_iterations == GOMEL_SCHEDULE_AUTO) fp64_iterations = gomel_auto_fp64_iterations(cfg, cfg->griffin_lim_iterations);
    const std::vector<size_t> b = plan_chunks(*cfg, n_utt, n_pairs, fp64_iterations != 0 || !gomel_supported_n(cfg->resolut));
    if (b.size() > cap) return GOMEL_ERR_BUFFER_TOO_SMALL;
    for (size_t i = 0; i < b.size(); i++) bounds[i] = b[i];
    *n_chunks = b.size() - 1;
    return GOMEL_OK;
    GOMEL_CATCH
}
int gomel_plan_chunks(const gomel_config* cfg, size_t n_utt, const size_t* n_pairs, size_t* bounds, size_t cap, size_t* n_chunks) {
    return gomel_plan_chunks_mode(cfg, GOMEL_SCHEDULE_AUTO, n_utt, n_pairs, bounds, cap, n_chunks);
}

// ---- diagnostics for the host-side logic (not in the header; tests/host_stress.cpp and the sanitizer runs reach them by name) ----
// A fake pin backend: registrations live in a map of its own; overlapping registrations and unknown unregistrations are VIOLATIONS of
// what the registry promises the runtime.
static std::mutex g_fake_mu;
static std::map<uintptr_t, size_t> g_fake_regs;
static std::atomic<long> g_fake_violations{0}, g_fake_calls{0};
static bool fake_pin_foreign(void*, uintptr_t* base, size_t* size) { *base = 0; *size = 0; return false; }
static bool fake_pin_reg(void* p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_fake_mu);
    g_fake_calls++;
    const uintptr_t a = (uintptr_t)p;
    auto it = g_fake_regs.upper_bound(a);
    if (it != g_fake_regs.begin()) { auto pr = std::prev(it); if (a < pr->first + pr->second) { g_fake_violations++; return false; } }
    if (it != g_fake_regs.end() && it->first < a + bytes) { g_fake_violations++; return false; }
    g_fake_regs[a] = bytes;
    return true;
}
static void fake_pin_unreg(void* p) {
    std::lock_guard<std::mutex> lk(g_fake_mu);
    g_fake_calls++;
    if (g_fake_regs.erase((uintptr_t)p) != 1) g_fake_violations++;
}
static const PinBackend kFakePins = {fake_pin_foreign, fake_pin_reg, fake_pin_unreg};
int gomel_debug_pin_backend(int fake) {   // only while nothing is registered
    std::lock_guard<std::mutex> lk(g_pins.mu);
    if (!g_pins.live.empty()) return GOMEL_ERR_BAD_CONFIG;
    g_pins.be = fake ? &kFakePins : g_pins.product_be;
    g_pins.foreign.clear();
    return GOMEL_OK;
}
int gomel_debug_pin_stats(long* live_registry, long* live_backend, long* violations, long* backend_calls) {
    { std::lock_guard<std::mutex> lk(g_pins.mu); if (live_registry) *live_registry = (long)g_pins.live.size(); }
    { std::lock_guard<std::mutex> lk(g_fake_mu); if (live_backend) *live_backend = (long)g_fake_regs.size(); }
    if (violations) *violations = g_fake_violations.load();
    if (backend_calls) *backend_calls = g_fake_calls.load();
    return GOMEL_OK;
}
// The per-call pin scope (PinnedScope: want / commit / release at scope exit) on caller buffers, as gomel_from_mel_batch uses it --
// without the copies: lets threads hammer the registry with aliased and nested ranges.
int gomel_debug_pin_scope(void* const* ptrs, const size_t* bytes, size_t n, int hold_us) {
    GOMEL_TRY
    PinWants pins;
    for (size_t i = 0; i < n; i++) pins.want(ptrs[i], bytes[i]);
    pins.commit();
    if (hold_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(hold_us));
    const int held = (int)pins.held.size();
    pins.release_all();
    return held;
    GOMEL_CATCH
}
// The device-buffer cache against host malloc: random allocate / release / trim sequences with its invariants checked
// (`cached` = the bytes of the unused blocks; a released block is handed out again or freed exactly once; clear() leaves nothing).
// Returns the number of violated invariants.
static thread_local long g_selftest_live = 0;   // (per thread: several self-tests may run at once)
int gomel_debug_pool_selftest(unsigned seed, int iters) {
    GOMEL_TRY
    DevPool pool;
    pool.alloc_fn = [](void** q, size_t n) { *q = malloc(n); if (*q) g_selftest_live++; return *q ? 0 : 2; };
    pool.free_fn = [](void* q) { free(q); g_selftest_live--; return 0; };
    pool.cap = (size_t)8 << 20;
    const long live0 = g_selftest_live;
    std::vector<std::pair<void*, size_t>> held;
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    int bad = 0;
    for (int i = 0; i < iters; i++) {
        const uint64_t r = rnd();
        if (held.size() < 24 && (r & 3) != 0) {
            void* p = nullptr;
            const size_t n = (size_t)1 << (8 + r % 13);
            if (pool.alloc(&p, n + (r >> 20) % 4096) != 0 || !p) { bad++; continue; }
            memset(p, 0xa5, 64);                                    // the block is really ours (ASan would object otherwise)
            for (auto& h : held) if (h.first == p) bad++;            // handed out twice
            held.emplace_back(p, n);
        } else if (!held.empty()) {
            const size_t k = r % held.size();
            pool.release(held[k].first);
            held.erase(held.begin() + k);
        }
        size_t unused = 0;
        for (const DevPool::Block& b : pool.blocks) if (!b.used) unused += b.bytes;
        if (unused != pool.cached || pool.cached > pool.cap) bad++;
        if ((r & 1023) == 0) { pool.trim(0); if (pool.cached != 0) bad++; }
    }
    for (auto& h : held) pool.release(h.first);
    pool.clear();
    if (g_selftest_live != live0) bad++;
    return bad;
    GOMEL_CATCH
}

// The caller's pointer -> the registration that covers it.  acquire() hands back the BASE of an enclosing live registration when the
// range lies inside one (a per-call pin of another thread, an earlier larger register), so the pointer itself is not the token.
static std::mutex g_reg_mu;
static std::multimap<uintptr_t, uintptr_t> g_reg_tokens;
int gomel_host_register(void* p, size_t bytes) {   // keeps the range pinned until gomel_host_unregister (reference-counted with the per-call pins)
    if (!p || bytes == 0) return GOMEL_ERR_NULL;
    GOMEL_TRY
    const uintptr_t tok = g_pins.acquire(p, bytes);
    if (!tok) return GOMEL_ERR_HIP;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_reg_tokens.emplace((uintptr_t)p, tok);
    return GOMEL_OK;
    GOMEL_CATCH
}
int gomel_host_unregister(void* p) {
    if (!p) return GOMEL_ERR_NULL;
    uintptr_t tok = 0;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        auto it = g_reg_tokens.find((uintptr_t)p);
        if (it == g_reg_tokens.end()) return GOMEL_ERR_BAD_CONFIG;   // not a pointer gomel_host_register took
        tok = it->second;
        g_reg_tokens.erase(it);
    }
    g_pins.release(tok);
    return GOMEL_OK;
}

int gomel_numa_cpus(const char* sysfs_root, const char* pci_bus_id, int32_t* node, int32_t* cpus, size_t cap, size_t* n_cpus) {
    GOMEL_TRY
    std::vector<int> v;
    int nd = -1;
    const int rc = numa_cpus_of(sysfs_root, pci_bus_id, &nd, v);
    if (node) *node = nd;
    if (n_cpus) *n_cpus = rc == GOMEL_OK ? v.size() : 0;
    if (rc) return rc;
    if (cpus) for (size_t i = 0; i < v.size() && i < cap; i++) cpus[i] = v[i];
    return GOMEL_OK;
    GOMEL_CATCH
}
int gomel_numa_bind_thread(const char* sysfs_root, const char* pci_bus_id, int32_t* node, int32_t* n_bound) {
    GOMEL_TRY
    if (n_bound) *n_bound = 0;
    std::vector<int> v;
    int nd = -1;
    const int rc = numa_cpus_of(sysfs_root, pci_bus_id, &nd, v);
    if (node) *node = nd;
    if (rc) return rc;
    cpu_set_t have, want;
    CPU_ZERO(&have); CPU_ZERO(&want);
    if (pthread_getaffinity_np(pthread_self(), sizeof have, &have) != 0) return GOMEL_ERR_EMPTY;
    int n = 0;
    for (int c : v) if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &have)) { CPU_SET(c, &want); n++; }
    if (n == 0) return GOMEL_ERR_EMPTY;                                         // the node's CPUs lie outside this thread's mask (cgroup, taskset): leave it alone
    if (pthread_setaffinity_np(pthread_self(), sizeof want, &want) != 0) return GOMEL_ERR_EMPTY;
    if (n_bound) *n_bound = n;
    return GOMEL_OK;
    GOMEL_CATCH
}

}  // extern "C"
