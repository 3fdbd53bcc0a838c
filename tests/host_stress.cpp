// host_stress.cpp -- threaded stress of libgomel_hip.so's HOST-side logic on a machine WITHOUT a GPU, meant to run under
// AddressSanitizer + UBSan and under ThreadSanitizer (scripts/sanitize_host.sh, tests/test_sanitizers.py; SURVEY section 5 "race
// detection / sanitizers": the reference is single-goroutine, the replacement's handle must be safe under concurrent cgo calls).
// Exercises, from several threads at once:
//   * the process-wide pin registry (reference-counted interval registry behind every per-call hipHostRegister) through
//     gomel_host_register / gomel_host_unregister and the per-call pin scope, on nested / aliased / disjoint ranges of shared
//     buffers -- against an in-process fake backend (gomel_debug_pin_backend) that counts violations of the runtime's contract
//     (overlapping registrations, unregistering what is not registered);
//   * the table builder (gomel_tables_build: Hann, undomel / domel rows, mel/impl.go:311-322,348-359) on random configurations,
//     every thread checking that the same configuration gives the same bytes;
//   * the chunk planner (gomel_plan_chunks) and the automatic precision schedule on random inputs (bounds strictly increasing, ...);
//   * the device-buffer cache against host malloc (gomel_debug_pool_selftest: allocate / release / trim invariants);
//   * the device gate (DeviceGateT of gomel_host.hpp, instantiated here on FAKE events): threads enter / leave as ordinary and as
//     persistent runs; a persistent run is never admitted while an ordinary one is enqueueing or in flight, the accounted share of the
//     device never exceeds 1024/1024 and returns to zero.
// Exit code 0 = no violation.  Diagnostic entry points are declared here by hand: they are not part of include/gomel_hip.h.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "gomel_hip.h"
#include "../gomel_amd/csrc/gomel_host.hpp"   // header-only, device-free: the gate template is instantiated below on fake events

extern "C" {
int gomel_debug_pin_backend(int fake);
int gomel_debug_pin_stats(long* live_registry, long* live_backend, long* violations, long* backend_calls);
int gomel_debug_pin_scope(void* const* ptrs, const size_t* bytes, size_t n, int hold_us);
int gomel_debug_pool_selftest(unsigned seed, int iters);
}

static std::atomic<long> g_fail{0};
#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) { g_fail++; fprintf(stderr, "host_stress: %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

struct Rng {
    uint64_t x;
    explicit Rng(uint64_t s) : x(s * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
    uint64_t next() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }
    uint64_t below(uint64_t n) { return next() % n; }
};

static const size_t kBuf = (size_t)48 << 20;   // the per-call scope pins ranges of 1 MiB and more
static char* g_buf[3];

static void pin_worker(int id, int iters) {
    Rng r(100 + id);
    for (int i = 0; i < iters; i++) {
        char* base = g_buf[r.below(3)];
        const uint64_t kind = r.below(4);
        if (kind == 0) {   // explicit register / unregister of a random sub-range (may nest inside somebody's live registration, may be refused)
            const size_t off = (size_t)r.below(kBuf - ((size_t)4 << 20)), len = ((size_t)1 << 20) + (size_t)r.below((size_t)3 << 20);
            const int rc = gomel_host_register(base + off, len);
            CHECK(rc == GOMEL_OK || rc == GOMEL_ERR_HIP);                      // refused = the range straddles a live registration
            if (rc == GOMEL_OK) {
                if (r.below(2)) std::this_thread::yield();
                CHECK(gomel_host_unregister(base + off) == GOMEL_OK);
            }
        } else if (kind == 1) {   // a pointer that was never registered
            CHECK(gomel_host_unregister(base + 7 + r.below(1000)) == GOMEL_ERR_BAD_CONFIG);
        } else {   // a call's scope: several buffers, prefix views of one base, repeated pointers
            void* ptrs[6]; size_t bytes[6];
            const size_t off = (size_t)r.below(kBuf / 2);
            for (int k = 0; k < 6; k++) {
                const bool same = k && r.below(3) == 0;
                ptrs[k] = same ? ptrs[k - 1] : base + ((off + (size_t)k * ((size_t)5 << 20)) % (kBuf - ((size_t)5 << 20)));
                bytes[k] = ((size_t)1 << 20) + (size_t)r.below((size_t)4 << 20);
            }
            const int held = gomel_debug_pin_scope(ptrs, bytes, 6, (int)r.below(30));
            CHECK(held >= 0 && held <= 6);
        }
    }
}

static void fill_config(Rng& r, gomel_config* c) {
    static const int ns[] = {256, 512, 1024, 2048, 4096, 1000, 3000, 63, 8192};
    gomel_default_config(c);
    c->resolut = ns[r.below(9)];
    c->window = 1 + (int)r.below((uint64_t)c->resolut);
    c->num_mels = 8 + (int)r.below(250);
    c->mel_fmax = 2000.0 + (double)r.below(20000);
    c->mel_fmin = (double)r.below(200);
    c->griffin_lim_iterations = (int)r.below(120);
    c->hann_periodic = (int)r.below(2);
}

static void table_worker(int id, int iters) {
    Rng r(7);   // every thread walks the SAME configurations: same bytes expected, and concurrent builds must not disturb each other
    std::vector<unsigned char> a, b;
    for (int i = 0; i < iters; i++) {
        gomel_config c;
        fill_config(r, &c);
        const size_t n = gomel_tables_bytes_for(&c);
        CHECK(n > 0);
        a.assign(n, 0xcc); b.assign(n, 0x33);
        const int ra = gomel_tables_build(&c, a.data(), n), rb = gomel_tables_build(&c, b.data(), n);
        CHECK(ra == rb && (ra == GOMEL_OK || ra == GOMEL_ERR_TABLE_RANGE));
        if (ra == GOMEL_OK) CHECK(memcmp(a.data(), b.data(), n) == 0);
        CHECK(gomel_tables_build(&c, a.data(), n - 1) == GOMEL_ERR_BUFFER_TOO_SMALL);
        const int k = gomel_auto_fp64_iterations(&c, c.griffin_lim_iterations);
        CHECK(k >= 0 && k <= c.griffin_lim_iterations);
        (void)id;
    }
}

static void plan_worker(int id, int iters) {
    Rng r(900 + id);
    for (int i = 0; i < iters; i++) {
        gomel_config c;
        fill_config(r, &c);
        const size_t n_utt = 1 + (size_t)r.below(700);
        std::vector<size_t> np(n_utt), bounds(70);
        for (size_t u = 0; u < n_utt; u++) np[u] = (size_t)c.num_mels * (1 + (size_t)r.below(900));
        size_t nch = 0;
        const int rc = gomel_plan_chunks(&c, n_utt, np.data(), bounds.data(), bounds.size(), &nch);
        CHECK(rc == GOMEL_OK);
        if (rc == GOMEL_OK) {
            CHECK(nch >= 1 && nch <= 65 && bounds[0] == 0 && bounds[nch] == n_utt);
            for (size_t k = 0; k < nch; k++) CHECK(bounds[k] < bounds[k + 1]);
        }
        np[r.below(n_utt)] += 1;                                              // no longer a multiple of NumMels
        CHECK(gomel_plan_chunks(&c, n_utt, np.data(), bounds.data(), bounds.size(), &nch) == GOMEL_ERR_BAD_LENGTH);
        CHECK(gomel_pad_len(1000 + (size_t)r.below(100000), c.window) >= 1000);
    }
}

// ---- the device gate on fake events: an event is "ready" once its flag is set ------------------------------------------------------
struct FakeEvent { std::atomic<int>* done; bool operator==(const FakeEvent& o) const { return done == o.done; } };
struct FakeEvOps {
    typedef FakeEvent Event;
    static int query(FakeEvent e) { return e.done->load() ? 0 : 1; }
};
typedef DeviceGateT<FakeEvOps> FakeGate;
static FakeGate g_fake_gate;
static std::atomic<int> g_ordinary_in_flight{0};   // ordinary runs between enter_ordinary and the completion of their event
static std::atomic<long> g_persist_admitted{0}, g_persist_refused{0}, g_waits_seen{0};
// Drives the PRODUCT's admission protocol (DeviceGateT::enter_ordinary / leave_ordinary / try_take_persistent / record_persistent /
// release_persistent / forget -- the very methods gomel_hip.hip's GateOrdinary and GatePersist call) from four threads, and asserts its
// invariant at the only place it can be observed without a race: under the gate's mutex, right after a persistent launch was admitted.
static void gate_worker(int id, int iters) {
    Rng r(5000 + id);
    std::vector<std::atomic<int>> flags(8);
    for (auto& f : flags) f = 1;
    int owner_tag = id;
    for (int i = 0; i < iters; i++) {
        std::atomic<int>& flag = flags[r.below(flags.size())];
        if (r.below(3) != 0) {   // an ordinary run: enter (told which persistent launches to wait for), enqueue, record its event, complete a little later
            g_fake_gate.enter_ordinary([](FakeEvent e) { g_waits_seen++; CHECK(e.done != nullptr); });
            g_ordinary_in_flight++;
            flag = 0;
            const FakeEvent ev{&flag};
            g_fake_gate.leave_ordinary(&ev, &owner_tag);
            if (r.below(2)) std::this_thread::yield();
            g_ordinary_in_flight--;
            flag = 1;             // (decrement first: an observer that sees the event done has seen the count lowered)
        } else {                 // a persistent run: admitted only while nothing ordinary is enqueueing or in flight and the share fits
            const int units = 128 + (int)r.below(900);
            if (!g_fake_gate.try_take_persistent(units, &owner_tag)) { g_persist_refused++; continue; }
            // admitted: the gate's mutex is held until record_persistent.  Every ordinary run that has entered has also left (enqueueing
            // == 0) and every event of another owner has completed -- and this worker's own runs completed before it came here.
            CHECK(g_ordinary_in_flight.load() == 0);
            CHECK(g_fake_gate.enqueueing == 0);
            CHECK(g_fake_gate.persist_units <= 1024);
            for (const FakeGate::Ent& e : g_fake_gate.np) CHECK(e.owner == &owner_tag);
            g_persist_admitted++;
            flag = 0;
            const FakeEvent ev{&flag};
            g_fake_gate.record_persistent(&ev, &owner_tag);
            if (r.below(2)) std::this_thread::yield();
            flag = 1;
            g_fake_gate.release_persistent(units, &owner_tag);
        }
    }
    g_fake_gate.forget(&owner_tag);   // this worker's flags die with it: nothing of its may stay in the gate
}

// ---- gomel_job_wait_timeout's polling loop on a fake event and a fake clock ---------------------------------------------------------
struct FakeClock {
    static int64_t& t() { static thread_local int64_t v = 0; return v; }
    static int& sleeps() { static thread_local int v = 0; return v; }
    static int64_t now_us() { return t(); }
    static void sleep_us(int64_t us) { CHECK(us > 0 && us <= 1000); t() += us; sleeps()++; }
};
struct CountdownEvOps {   // "done" after a number of queries; a negative count makes the query fail
    typedef int* Event;
    static int query(int* left) { if (*left < 0) return -1; if (*left == 0) return 0; (*left)--; return 1; }
};
static void bounded_wait_checks() {
    int left = 0;
    FakeClock::t() = 0; FakeClock::sleeps() = 0;
    CHECK((wait_event_bounded<CountdownEvOps, FakeClock>(&left, 1000) == 0) && FakeClock::sleeps() == 0);   // already done: no sleep
    left = 5;
    CHECK((wait_event_bounded<CountdownEvOps, FakeClock>(&left, 1000) == 0) && FakeClock::sleeps() == 5);   // done after five polls
    left = 1 << 30; FakeClock::t() = 0; FakeClock::sleeps() = 0;
    CHECK((wait_event_bounded<CountdownEvOps, FakeClock>(&left, 20) == 1));                                  // still running at the deadline
    CHECK(FakeClock::t() == 20000 && FakeClock::sleeps() >= 20 && FakeClock::sleeps() < 40);                  // slept exactly up to it, in naps of at most 1 ms
    left = 1 << 30; FakeClock::sleeps() = 0;
    CHECK((wait_event_bounded<CountdownEvOps, FakeClock>(&left, 0) == 1) && FakeClock::sleeps() == 0);       // timeout 0 = one poll
    CHECK((wait_event_bounded<CountdownEvOps, FakeClock>(&left, -5) == 1) && FakeClock::sleeps() == 0);      // negative = the same
    left = -1;
    CHECK((wait_event_bounded<CountdownEvOps, FakeClock>(&left, 1000) == -1));                               // a failing query ends the wait
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    for (char*& b : g_buf) { b = (char*)malloc(kBuf); if (!b) return 2; memset(b, 0, kBuf); }
    CHECK(gomel_debug_pin_backend(1) == GOMEL_OK);
    std::vector<std::thread> th;
    for (int i = 0; i < 6; i++) th.emplace_back(pin_worker, i, iters * 4);
    for (int i = 0; i < 3; i++) th.emplace_back(table_worker, i, iters / 6 + 1);
    for (int i = 0; i < 3; i++) th.emplace_back(plan_worker, i, iters);
    std::atomic<int> pool_bad{0};
    for (int i = 0; i < 2; i++) th.emplace_back([&pool_bad, i, iters] { pool_bad += gomel_debug_pool_selftest(11u + (unsigned)i, iters * 20); });
    for (int i = 0; i < 4; i++) th.emplace_back(gate_worker, i, iters * 20);
    for (std::thread& t : th) t.join();
    long live_reg = -1, live_be = -1, viol = -1, calls = -1;
    gomel_debug_pin_stats(&live_reg, &live_be, &viol, &calls);
    CHECK(live_reg == 0);     // every acquire was released
    CHECK(live_be == 0);      // ... and reached the backend as an unregister
    CHECK(viol == 0);         // the registry never asked the runtime for an overlapping registration / an unknown unregistration
    CHECK(calls > 0);
    CHECK(pool_bad.load() == 0);
    CHECK(g_fake_gate.persist_units == 0 && g_fake_gate.enqueueing == 0 && g_fake_gate.np.empty() && g_fake_gate.ps.empty());
    CHECK(g_persist_admitted.load() > 0 && g_persist_refused.load() > 0);   // both outcomes of the admission were exercised
    bounded_wait_checks();
    CHECK(gomel_debug_pin_backend(0) == GOMEL_OK);
    for (char* b : g_buf) free(b);
    printf("host_stress: %ld failures; pin backend calls %ld, violations %ld, live %ld/%ld; pool self-test violations %d; gate: %ld persistent launches admitted, %ld refused, %ld waits handed out\n",
           g_fail.load(), calls, viol, live_reg, live_be, pool_bad.load(), g_persist_admitted.load(), g_persist_refused.load(), g_waits_seen.load());
    return g_fail.load() ? 1 : 0;
}
