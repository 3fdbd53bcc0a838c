// gl_pipe.hpp -- k_gl_fused_d2: the fp64 fused Griffin-Lim kernel (k_gl_fused_d, gl_kernels.hpp) with TWO FRAMES IN FLIGHT per stream.
//
// Replaces, like k_gl_fused_d, one iteration of the loop of ISTFT (mel/mel.go:85-136) over a whole batch: frame * Hann -> fft.FFTReal ->
// magnitude clamp / phase keep -> fft.IFFT -> * Hann -> overlap-add.  SAME arithmetic per frame, same summation order of the overlap-add,
// same segments and tails: its output equals k_gl_fused_d's BIT FOR BIT (tests/test_gpu_round6.py) -- only the schedule differs.
//
// Why: k_gl_fused_d runs one wave per SIMD (376 registers, 152 KB of LDS per workgroup), so nothing overlaps: of ~14 300 cycles per frame
// and wave ~9 600 are instruction issue and ~4 500 are LDS round trips the wave sits out -- four exchanges per frame, each "write burst ->
// partner wave's event -> read burst -> data" (DESIGN.md 4.2b).  Frames of one iteration are independent (they only meet in the overlap-add),
// so here a stream software-pipelines its segment: while frame g goes through its INVERSE FFT (role A), frame g+1 goes through its FORWARD
// FFT (role B), pass by pass, each role in an exchange area of its own:
//
//     A: pass-3 butterflies + write | B: pass-1 butterfly + twiddles       (B's arithmetic covers A's write -> partner -> read round trip)
//     A: wait, issue reads          | B: wait, write
//     A: twiddles + pass-2 butterfly| B: wait, issue reads                  (A's arithmetic covers B's)
//     A: wait, write                | B: pass-2 butterfly + twiddles
//     A: wait, issue reads          | B: wait, write
//     A: twiddles + pass-1 butterfly| B: wait, issue reads
//     A: * window, overlap-add, stores (frame g is done)                    (covers B's last read)
//                                   | B: pass-3 butterflies, spectral step  (frame g+1 is now what role A takes up in the next round)
//
// What it costs: two exchange areas per stream (4 x 34.8 KB per workgroup) leave no room in the 160 KB LDS for the pass-1 twiddles (32 KB) and the
// spectral step's W_N^k (16 KB): both are lane constants and live in registers for the whole launch (the compiler parks them in AGPRs:
// ~150 v_accvgpr_read per frame); the window table shrinks to HALF -- window[n] == window[N-1-n], which gossp's Hann (the only window the
// reference ever uses, mel/mel.go:50,145) satisfies; the host checks the table's two halves bit for bit (gomel_create / gomel_set_window) and
// keeps k_gl_fused_d for any other window.  Built for N = 4096 (two waves per stream: the kernel the default call spends 75 % of its time in).
#pragma once
#include <type_traits>

namespace glpipe {
using namespace glfft;

// ---- role A: the inverse FFT of one frame, in pieces (the arithmetic of fft_inverse, gl_fft.hpp, statement for statement)
template <class G> GL_DEV void inv_p3_write(v2d (&v)[16], v2d* lds, int t) {   // pass-3 butterflies; the first one's outputs leave while the second computes
    const int bA = t, bB = t ? G::B3 - t : G::B3 / 2;
    dft8<true, 1>(&v[0]);
#pragma unroll
    for (int r = 0; r < 8; r++) lds[r * G::SB + bA] = v[r];
    dft8<true, 1>(&v[8]);
#pragma unroll
    for (int r = 0; r < 8; r++) lds[r * G::SB + bB] = v[8 + r];
    __builtin_amdgcn_sched_group_barrier(GL_SGB_VALU, 56, 0);
    sgb_valu_then_write<7, 8>();
    sgb_valu_then_write<0, 8>();
}
template <class G> GL_DEV void inv_p2_read(v2d (&v)[16], v2d (&w2)[16], const v2d* lds, const v2d* tw2, int t) {
    static_assert(G::NB2 == 1, "one pass-2 butterfly per lane (N 4096, 2048)");
    const int q1 = t % G::R1, t2 = t / G::R1;
#pragma unroll
    for (int q = 0; q < G::R2; q++) {
        v[q] = lds[t2 * G::SB + q1 + G::R1 * q];
        if (q) w2[q] = tw2[q * 8 + t2];
    }
}
template <class G> GL_DEV void inv_p2_compute(v2d (&v)[16], const v2d (&w2)[16]) {
    apply_tw<true, 1, G::R2>(&v[0], w2);
    dftR<true, G::R2, 1>(&v[0]);
}
template <class G> GL_DEV void inv_p2_write(const v2d (&v)[16], v2d* lds, int t) {
    const int q1 = t % G::R1, t2 = t / G::R1;
#pragma unroll
    for (int m = 0; m < G::R2; m++) lds[q1 * G::SAI + t2 + 8 * m] = v[m];
}
template <class G> GL_DEV void inv_p1_read(v2d (&v)[16], const v2d* lds, int t) {
    static_assert(G::NB1 == 1, "one pass-1 butterfly per lane (N 4096)");
#pragma unroll
    for (int q = 0; q < G::R1; q++) v[q] = lds[q * G::SAI + t];
}
template <class G> GL_DEV void inv_p1_compute(v2d (&v)[16], const v2d (&w1)[16]) {
    apply_tw<true, 1, G::R1>(&v[0], w1);
    dftR<true, G::R1, 1>(&v[0]);
}

// ---- role B: the forward FFT of one frame, in pieces (fft_forward's interleaved form)
template <class G> GL_DEV void fwd_p1_compute(v2d (&v)[16], const v2d (&w1)[16]) {
    dftR<false, G::R1, 1>(&v[0]);
#pragma unroll
    for (int q = 1; q < G::R1; q++) v[q] = cmul(v[q], w1[q]);
}
template <class G> GL_DEV void fwd_p1_write(const v2d (&v)[16], v2d* lds, int t) {
#pragma unroll
    for (int q = 0; q < G::R1; q++) lds[q * G::SA + t] = v[q];
}
template <class G> GL_DEV void fwd_p2_read(v2d (&v)[16], v2d (&w2)[16], const v2d* lds, const v2d* tw2, int t) {
    const int q1 = t % G::R1, t2 = t / G::R1;
#pragma unroll
    for (int m = 0; m < G::R2; m++) v[m] = lds[q1 * G::SA + t2 + 8 * m];
#pragma unroll
    for (int q = 1; q < G::R2; q++) w2[q] = tw2[q * 8 + t2];
}
template <class G> GL_DEV void fwd_p2_compute(v2d (&v)[16], const v2d (&w2)[16]) {
    dftR<false, G::R2, 1>(&v[0]);
#pragma unroll
    for (int q = 1; q < G::R2; q++) v[q] = cmul(v[q], w2[q]);
}
template <class G> GL_DEV void fwd_p2_write(const v2d (&v)[16], v2d* lds, int t) {
    const int q1 = t % G::R1, t2 = t / G::R1;
#pragma unroll
    for (int q = 0; q < G::R2; q++) lds[t2 * G::SB + q1 + G::R1 * q] = v[q];
}
template <class G> GL_DEV void fwd_p3_read(v2d (&v)[16], const v2d* lds, int t) {
    const int bA = t, bB = t ? G::B3 - t : G::B3 / 2;
#pragma unroll
    for (int r = 0; r < 8; r++) { v[r] = lds[r * G::SB + bA]; v[8 + r] = lds[r * G::SB + bB]; }
}
GL_DEV void fwd_p3_compute(v2d (&v)[16]) {
    dft8<false, 1>(&v[0]);
    dft8<false, 1>(&v[8]);
}

// A lane's 16 window pairs from the HALF table (pairs 0 .. C/2-1): rows j < 8 directly (pair t + T j), rows j >= 8 are the mirror images
// of rows 15-j of lane T-1-t with the two samples swapped -- window[N-1-n] == window[n].
template <class G> GL_DEV void load_window_half(v2d (&w)[16], const v2d* winh, int t) {
    constexpr int T = G::T;
    static_assert(G::C == 16 * T, "16 pairs per lane");
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = winh[t + T * j];
#pragma unroll
    for (int j = 8; j < 16; j++) { const v2d m = winh[(T - 1 - t) + T * (15 - j)]; w[j] = v2d{m.y, m.x}; }
}

#define GL_PIPE_FENCE() __builtin_amdgcn_sched_barrier(0)
// The values of a role are COMPUTED where the schedule says, not where their first use is: without this the optimizer sinks a phase's
// arithmetic below the next poll, next to the LDS writes that consume it -- exactly the serialisation the pipeline is there to remove.
GL_DEV void pin(v2d (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 16; j++) asm volatile("" : "+v"(v[j]));
}
#define GL_SGB_DSR 0x100

// One poll for the partner events of both roles: two LDS reads, one wait (a poll drains the wave's LDS queue, so the fewer the better).
// Bounded like StreamFlagSync::wait: the first wait to run out of polls raises the error word, later ones give up after one poll.
GL_DEV void wait_both(StreamFlagSync<true>& a, StreamFlagSync<true>& b) {
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    asm volatile("" ::: "memory");
    unsigned polls = 0;
    for (;;) {
        const unsigned fa = *(volatile lds_u32*)a.other, fb = *(volatile lds_u32*)b.other;
        const unsigned ra = __builtin_amdgcn_readfirstlane(fa), rb = __builtin_amdgcn_readfirstlane(fb);   // BOTH consumed on every path: a flag read left pending
        if ((ra >= a.ev) & (rb >= b.ev)) break;                                                            // at the loop's exit costs an lgkmcnt(0) at the next use of its register
        if (__builtin_expect(++polls > a.cap, 0)) {
            a.cap = 0;
            if (a.err) *(volatile __attribute__((address_space(1))) unsigned*)a.err = 1u;   // (a GLOBAL store: a flat one pending at the join poisons every counted LDS wait behind the poll)
            break;
        }
    }
    asm volatile("" ::: "memory");
}

template <int N> static constexpr size_t fused_d2_lds_bytes() {
    typedef Geo<N> G;
    return ((size_t)(kFusedThreads / G::T) * 2 * G::LDS_SLOTS + G::TW2_N + G::C / 2) * 16 + (size_t)(kFusedThreads / G::T) * 16 + 16;   // areas | TW2 | half window | counters [STREAMS][2 roles][2 waves]
}

template <int N, int SH>
__global__ __launch_bounds__(kFusedThreads, 1) void k_gl_fused_d2(const double* __restrict__ x_in, const double* __restrict__ tail_in,
                                                                 double* __restrict__ x_out, double* __restrict__ tail_out,
                                                                 const SegDesc* __restrict__ segs, int n_segs, int g_max_unused,
                                                                 const double* __restrict__ mag, const double* __restrict__ nyq, DevTables tb) {
    typedef Geo<N> G;
    constexpr int T = G::T, STREAMS = kFusedThreads / T, KEEP = 16 - SH, TAIL_PAIRS = KEEP * T;
    static_assert(T == 128 && G::NB1 == 1 && G::NB2 == 1, "built for N = 4096: two waves per stream, one butterfly per lane and pass");
    extern __shared__ __attribute__((aligned(16))) char fused_d2_smem[];
    v2d* lds = reinterpret_cast<v2d*>(fused_d2_smem);             // [STREAMS][2 roles][LDS_SLOTS] | TW2 | half window | counters
    v2d* tw2_l = lds + STREAMS * 2 * G::LDS_SLOTS;
    v2d* winh_l = tw2_l + G::TW2_N;
    unsigned* cnt_sh = reinterpret_cast<unsigned*>(winh_l + G::C / 2);   // [STREAMS][2 roles][2 waves]
    if (threadIdx.x < 4 * STREAMS) cnt_sh[threadIdx.x] = 0u;
    for (int i = threadIdx.x; i < G::TW2_N; i += kFusedThreads) tw2_l[i] = tb.tw2d[i];
    for (int i = threadIdx.x; i < G::C / 2; i += kFusedThreads) winh_l[i] = reinterpret_cast<const v2d*>(tb.windowd)[i];
    __syncthreads();
    const int stream = threadIdx.x / T, t = threadIdx.x % T, wave = t >> 6;
    v2d* LA = lds + (stream * 2 + 0) * G::LDS_SLOTS;              // role A: the inverse FFT's exchange area
    v2d* LB = lds + (stream * 2 + 1) * G::LDS_SLOTS;              // role B: the forward FFT's
    unsigned evA = 0, evB = 0, sync_cap = tb.sync_cap;
    StreamFlagSync<true> syncA{cnt_sh + 4 * stream + wave, cnt_sh + 4 * stream + (wave ^ 1), evA, sync_cap, tb.sync_err};
    StreamFlagSync<true> syncB{cnt_sh + 4 * stream + 2 + wave, cnt_sh + 4 * stream + 2 + (wave ^ 1), evB, sync_cap, tb.sync_err};
    int seg = blockIdx.x * STREAMS + stream;
    const bool live = seg < n_segs;
    if (!live) seg = n_segs - 1;
    SegDesc d = segs[seg];
    d.G = __builtin_amdgcn_readfirstlane(d.G);
    d.has_prev = __builtin_amdgcn_readfirstlane(d.has_prev);
    d.has_next = __builtin_amdgcn_readfirstlane(d.has_next);
    seg = __builtin_amdgcn_readfirstlane(seg);
    const int seg_prev = d.has_prev ? seg - 1 : seg;
    const v2d* xin_p = reinterpret_cast<const v2d*>(x_in) + uniform_i64(d.s0 / 2);
    v2d* xout_p = reinterpret_cast<v2d*>(x_out) + uniform_i64(d.s0 / 2);
    const v2d* tprev = reinterpret_cast<const v2d*>(tail_in) + uniform_i64((int64_t)seg_prev * TAIL_PAIRS);
    const v2d* town = reinterpret_cast<const v2d*>(tail_in) + uniform_i64((int64_t)seg * TAIL_PAIRS);
    v2d* tout = reinterpret_cast<v2d*>(tail_out) + uniform_i64((int64_t)seg * TAIL_PAIRS);
    const double* mag_p = mag + uniform_i64(d.frame0 * G::C);
    const double* nyq_p = nyq + uniform_i64(d.frame0);
    const double f_prev = d.has_prev ? 1.0 : 0.0, f_next = d.has_next ? 1.0 : 0.0;
    const int rows_owned = d.G * SH;

    // Lane constants that no longer fit the LDS.  The eight W_N^k of the spectral step stay in registers for the whole launch.  The 15 pass-1
    // twiddles W_C^{t q} are fetched from the table in global memory (32 KB, the same for every workgroup: L2 hits) ONCE per round, in its
    // middle -- the inverse FFT's last pass and the next round's forward first pass use them, and between those two lies the quiet end of a
    // round; across the crowded first half of a round they are not live.  (Left to the register allocator they were spilled to scratch and
    // reloaded twice per round.)
    v2d w1[16], wn[8];
    w1[0] = v2d{1.0, 0.0};
    // (addresses as "uniform base + 32-bit lane offset": the loads take an SGPR base per row and ONE offset register -- fifteen hoisted 64-bit
    // lane addresses were what the allocator spilled, each reload with a full vmcnt(0) in front of its load)
    const char* tw1_base = reinterpret_cast<const char*>(tb.tw1d);
    const unsigned lane_off16 = (unsigned)t * 16u;
    auto load_w1 = [&]() {
        unsigned off = lane_off16;
        asm volatile("" : "+v"(off));   // the address is formed HERE, not hoisted out of the frame loop as fifteen 64-bit values
#pragma unroll
        for (int q = 1; q < G::R1; q++) w1[q] = *reinterpret_cast<const v2d*>(tw1_base + (size_t)q * G::C1 * 16 + off);
    };
    load_w1();
#pragma unroll
    for (int s8 = 0; s8 < 8; s8++) wn[s8] = tb.wnd[slot_bin<G>(t, s8)];
    const Twiddles<v2d> tw{tb.tw1d, tw2_l, nullptr, w1, G::R1 - 1, nullptr};

    v2d xin[16], acc[KEEP];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        v2d a = xin_p[T * j + t];
        if (j < KEEP) a += tprev[T * j + t] * f_prev;
        const int rn = j >= rows_owned ? j - rows_owned : 0;
        a += town[T * rn + t] * (j >= rows_owned ? f_next : 0.0);
        xin[j] = a;
    }
#pragma unroll
    for (int j = 0; j < KEEP; j++) acc[j] = v2d{0.0, 0.0};

    // "B input" of frame f: v = window * xin (xin holds frame f's 16 rows), then xin slides by SH rows
    v2d xl[SH], tl[SH];
    double fl[SH];
    bool any_next = false;
    MagRawD mraw;
    auto b_input = [&](v2d (&v)[16]) {
        v2d w[16];
        load_window_half<G>(w, winh_l, t);
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = w[j] * xin[j];
#pragma unroll
        for (int j = 0; j < KEEP; j++) xin[j] = xin[j + SH];
    };
    // what frame f still needs from memory: its magnitude row (spectral step) and the SH rows frame f+1 adds to the window.  Requested in the
    // MIDDLE of a round (two FFT passes before their first use) rather than at its top: 60-90 registers less across the round's first half
    auto b_requests = [&](int f) {
        load_mag_rows_d<G>(mraw, t, mag_p + (int64_t)f * G::C);
        mraw.nq = nyq_p[f];
        const bool more = f + 1 < d.G;
        const int row0 = (f + 1) * SH + KEEP;
        any_next = more && d.has_next && row0 + SH - 1 >= rows_owned;
#pragma unroll
        for (int jj = 0; jj < SH; jj++) xl[jj] = xin_p[T * (more ? row0 + jj : 0) + t];
        if (any_next) {
#pragma unroll
            for (int jj = 0; jj < SH; jj++) {
                const bool in_next = row0 + jj >= rows_owned;
                tl[jj] = town[T * (in_next ? row0 + jj - rows_owned : 0) + t];
                fl[jj] = in_next ? 1.0 : 0.0;
            }
        }
    };
    auto b_rows_in = [&]() {   // the requested rows complete xin for the next frame
#pragma unroll
        for (int jj = 0; jj < SH; jj++) xin[KEEP + jj] = xl[jj];
        if (any_next) {
#pragma unroll
            for (int jj = 0; jj < SH; jj++) xin[KEEP + jj] += tl[jj] * fl[jj];
        }
    };
    auto b_spectral = [&](v2d (&v)[16]) {
        MagRegsD mr;
        finish_mags_d(mr, mraw, t == 0);
        spectral_apply<G>(v, t, mr, wn);
    };
    auto a_ola = [&](const v2d (&v)[16], int g, auto last) {   // frame g's inverse FFT output: * window, overlap-add, the SH finished rows leave
        v2d w[16];
        load_window_half<G>(w, winh_l, t);
        if (live) {
            v2d* xo = xout_p + T * (g * SH);
#pragma unroll
            for (int j = 0; j < SH; j++) {
                const v2d y = v[j] * w[j];
                xo[T * j + t] = (j < KEEP) ? acc[j] + y : y;
            }
            v2d nacc[KEEP];
#pragma unroll
            for (int j = 0; j < KEEP; j++) {
                const v2d y = v[j + SH] * w[j + SH];
                nacc[j] = (j + SH < KEEP) ? acc[j + SH] + y : y;
            }
#pragma unroll
            for (int j = 0; j < KEEP; j++) acc[j] = nacc[j];
            if constexpr (decltype(last)::value) {   // the segment's last frame: what is left lies past the owned range
                if (d.has_next) {
#pragma unroll
                    for (int j = 0; j < KEEP; j++) tout[T * j + t] = acc[j];
                } else {
#pragma unroll
                    for (int j = 0; j < KEEP; j++) xout_p[T * (rows_owned + j) + t] = acc[j];
                }
            }
        }
    };

    // ---- prologue: frame 0 through its forward FFT and spectral step alone (role B's area and events: the same event sequence as a round's)
    v2d vA[16], vB[16];
    b_input(vB);
    b_requests(0);
    fft_forward<G, 0>(vB, LB, t, tw, syncB);
    b_spectral(vB);
    b_rows_in();

    // ---- rounds: frame g through its inverse FFT (role A) beside frame g+1 through its forward FFT (role B)
    // A role's exchange is "write burst -> partner's event -> read burst -> data": the OTHER role's arithmetic is what has to stand between those
    // steps.  So every butterfly is cut in two (butterfly | twiddles) and the other role's poll + read burst goes into the cut:
    //     C_A3+W_A3 | C_B1.a [poll, R_A2] C_B1.b+W_B1 | C_A2.a [poll, R_B2] C_A2.b+W_A2 | C_B2.a [poll, R_A1] C_B2.b+W_B2 | C_A1.a [poll, R_B3] C_A1.b, OLA_A, C_B3
    // (C = arithmetic, W = 16 LDS writes spread through it, R = 16 LDS reads; one poll checks the partner's events of BOTH roles, and it sits
    // where a poll must: in front of the burst it guards, after arithmetic that gave the partner time to get there.)
    for (int g = 0; g + 1 < d.G; g++) {
#pragma unroll
        for (int j = 0; j < 16; j++) vA[j] = vB[j];                 // frame g: from role B (spectrum Y_g) to role A
        v2d w2a[16], w2b[16];
        b_input(vB);
        pin(vB);
        GL_PIPE_FENCE();
        // S1: A pass 3 (the partner's last reads of role A's area were checked by the previous round's last poll / never happened)
        inv_p3_write<G>(vA, LA, t);
        syncA.arrive();
        GL_PIPE_FENCE();
        // S2: B pass 1 around A's pass-2 reads
        dftR<false, G::R1, 1>(&vB[0]);
        pin(vB);
        GL_PIPE_FENCE();
        wait_both(syncA, syncB);                                 // A's area written by the partner; B's area read by the partner (previous round)
        inv_p2_read<G>(vA, w2a, LA, tw2_l, t);
        syncA.arrive();
#pragma unroll
        for (int q = 0; q < G::R1; q++) {
            if (q) vB[q] = cmul(vB[q], w1[q]);
            LB[q * G::SA + t] = vB[q];
        }
        sgb_valu_then_write<4, 16>();
        syncB.arrive();
        GL_PIPE_FENCE();
        // S3: A pass 2 around B's pass-2 reads
        apply_tw<true, 1, G::R2>(&vA[0], w2a);
        pin(vA);
        GL_PIPE_FENCE();
        wait_both(syncA, syncB);                                 // B's area written; A's area read
        load_w1();                                               // (L2 hits: asked for first, so that the wait for them does not include the HBM loads below)
        b_requests(g + 1);                                       // (global memory: two passes ahead of their use)
        fwd_p2_read<G>(vB, w2b, LB, tw2_l, t);
        syncB.arrive();
        dftR<true, G::R2, 1>(&vA[0]);
        inv_p2_write<G>(vA, LA, t);
        __builtin_amdgcn_sched_group_barrier(GL_SGB_VALU, 64, 0);
        sgb_valu_then_write<3, 16>();
        syncA.arrive();
        GL_PIPE_FENCE();
        // S4: B pass 2 around A's pass-1 reads
        dftR<false, G::R2, 1>(&vB[0]);
        pin(vB);
        GL_PIPE_FENCE();
        wait_both(syncA, syncB);                                 // A's area written; B's area read
        inv_p1_read<G>(vA, LA, t);
        syncA.arrive();
        {
            const int q1 = t % G::R1, t2 = t / G::R1;
#pragma unroll
            for (int q = 0; q < G::R2; q++) {
                if (q) vB[q] = cmul(vB[q], w2b[q]);
                LB[t2 * G::SB + q1 + G::R1 * q] = vB[q];
            }
            sgb_valu_then_write<4, 16>();
        }
        syncB.arrive();
        GL_PIPE_FENCE();
        // S5: A pass 1 around B's pass-3 reads; then frame g leaves (window, overlap-add, stores) and frame g+1 finishes its forward side
        apply_tw<true, 1, G::R1>(&vA[0], w1);
        pin(vA);
        GL_PIPE_FENCE();
        wait_both(syncA, syncB);                                 // B's area written; A's area read (the next round's S1 writes it)
        fwd_p3_read<G>(vB, LB, t);
        syncB.arrive();
        dftR<true, G::R1, 1>(&vA[0]);
        pin(vA);
        GL_PIPE_FENCE();
        a_ola(vA, g, std::false_type{});
        GL_PIPE_FENCE();
        fwd_p3_compute(vB);
        b_spectral(vB);
        b_rows_in();
    }
    // ---- epilogue: the segment's last frame through its inverse FFT alone
#pragma unroll
    for (int j = 0; j < 16; j++) vA[j] = vB[j];
    fft_inverse<G, 0>(vA, LA, t, tw, syncA);
    a_ola(vA, d.G - 1, std::true_type{});
}

}  // namespace glpipe
