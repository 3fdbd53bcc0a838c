/* gomel_hip.h -- C ABI of libgomel_hip.so: the MI355X (gfx950) Griffin-Lim mel-inversion engine that sits
 * behind gomel's Go package API `github.com/neurlang/gomel/mel`.
 *
 * The reference has no FFI today (pure Go).  These entry points are what a cgo shim for the hot path binds
 * (bindings/go/mel/mel_hip.go).  Each one names the reference interface it replaces
 * (file:line relative to the reference repository).  Plain pointers and sizes only; every function returns a
 * gomel_status (0 = ok) and never throws or aborts across the boundary.  Conditions on which the Go code
 * panics (empty input, len % NumMels != 0, mel table running past the slice) come back as status codes.
 *
 * Layouts: a Go `[][2]float64` is one contiguous `double[2*n]`, frame-major: element (f*NumMels + k)*2 + c is
 * channel c of mel band k of frame f (producer domel mel/impl.go:313-341, consumer undomel mel/impl.go:350).
 * Waveforms are `double[L]`, L = Resolut + (F-1)*Window (mel/mel.go:80).
 *
 * Pointers: `mel_pairs`, `x0` and `out` of gomel_from_mel, gomel_from_mel_batch, gomel_batch_upload and
 * gomel_batch_download may each be host memory (pageable or pinned) or memory of the handle's device (a pipeline that
 * keeps its spectrograms in HBM hands over device pointers; the copies are hipMemcpyDefault).  Everything else is host.
 *
 * Threading: a handle may be used from concurrent host threads; calls on one handle are serialised by an
 * internal mutex; every entry switches the calling thread to the handle's HIP device and restores the thread's previous
 * device before it returns (a process that drives several GPUs finds its current device unchanged).
 */
#ifndef GOMEL_HIP_H
#define GOMEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOMEL_HIP_ABI_VERSION 6

typedef enum gomel_status {
    GOMEL_OK = 0,
    GOMEL_ERR_BAD_CONFIG = 1,     /* NumMels/Window/Resolut <= 0, TuneMul == 0, fmax+fmin <= 0 ...                 */
    GOMEL_ERR_UNSUPPORTED_N = 2,  /* Resolut outside what the FFT cores cover (see gomel_create)                   */
    GOMEL_ERR_BAD_LENGTH = 3,     /* n_pairs % NumMels != 0 (reference: index panic, mel/impl.go:366)             */
    GOMEL_ERR_EMPTY = 4,          /* zero frames (reference: index panic at mel/mel.go:78)                        */
    GOMEL_ERR_TABLE_RANGE = 5,    /* undomel/domel table indexes past the band/bin slice (reference: panic)       */
    GOMEL_ERR_BUFFER_TOO_SMALL = 6,
    GOMEL_ERR_NULL = 7,
    GOMEL_ERR_HIP = 8,            /* a HIP runtime call failed; gomel_last_hip_error() has the hipError_t         */
    GOMEL_ERR_NO_DEVICE = 9,
    GOMEL_ERR_BAD_TABLES = 10,    /* injected table blob does not match the configuration                         */
    GOMEL_ERR_TOO_SHORT = 11,     /* ToMel: padded input shorter than one frame                                   */
    GOMEL_ERR_NO_MEMORY = 12,     /* host allocation failed (C++ exceptions never cross the boundary)             */
    GOMEL_ERR_BUSY = 13           /* gomel_destroy with submitted jobs not yet waited for (the handle stays valid);
                                     gomel_job_wait_timeout: the job is still running at the deadline (it stays valid) */
} gomel_status;

/* Mirrors `type Mel` (mel/mel.go:10-27).  YReverse, VolumeBoost and SampleRate only matter to the PNG/WAV
 * codec, which stays on the host side of the boundary (ToWavPng mel/mel.go:211-238). */
typedef struct gomel_config {
    int32_t num_mels;               /* NumMels                                   default 160  */
    int32_t window;                 /* Window  = hop (gossp frameShift), any > 0 default 256  */
    int32_t resolut;                /* Resolut = frame / FFT length N            default 2048 */
    int32_t griffin_lim_iterations; /* GriffinLimIterations                      default 2    */
    double mel_fmin;                /* MelFmin  default 0    */
    double mel_fmax;                /* MelFmax  default 8000 */
    double tune_mul;                /* TuneMul  default 1    */
    double tune_add;                /* TuneAdd  default 0    */
    int32_t device;                 /* HIP device ordinal; -1 = the calling thread's current device */
    int32_t hann_periodic;          /* 0 = symmetric Hann 0.5-0.5cos(2 pi n/(N-1)) (gossp window.CreateHanning, assumed);
                                       1 = periodic (2 pi n/N).  gomel_set_window() injects any other table. */
} gomel_config;

typedef struct gomel_handle gomel_handle; /* configuration + device tables + stream       */
typedef struct gomel_batch gomel_batch;   /* device-resident batch of utterances           */

/* Per-run timing, measured with HIP events on the handle's stream. */
typedef struct gomel_timing {
    float total_ms;        /* mel->magnitude + x0 init + all iterations + finalize                        */
    float iter_ms;         /* the Griffin-Lim iteration launches only (sum over iterations)               */
    float dom_ms;          /* the dominant kernel alone, summed over its launches (events around each)    */
    int32_t iter_launches; /* number of dominant-kernel launches                                          */
    int32_t fast_path;     /* 0 = generic frames + overlap-add kernels, 1 = fused sliding-window kernel, 2 = persistent kernel,
                              3 = fused kernel, any-even-hop form */
    int64_t frames;        /* frames processed                                                            */
    float dom64_ms;        /* of dom_ms: the launches of the fp64 fused kernel (64-bit mode, precision schedule) */
    int32_t iter64_launches; /* how many of iter_launches those were                                      */
} gomel_timing;

/* NewMel() defaults, mel/mel.go:30-41 (device = -1, symmetric Hann). */
int gomel_default_config(gomel_config* cfg);

/* Replaces stft.New(m.Window, m.Resolut) (mel/mel.go:50,145) and the per-call index math of undomel/domel
 * (mel/impl.go:311-322,348-359): builds Hann, FFT twiddles and the mel<->linear tables once, uploads them.
 * Resolut: 256, 512, 1024, 2048, 4096 run the tuned register FFT cores; every other length from 4 to 65536 (odd lengths
 * included: the reference's fft.FFTReal / fft.IFFT take any length, mel/mel.go:95,116) runs generic fp64 kernels -- up to 8192 with
 * the frame in LDS (radix-2 for powers of two, Bluestein otherwise), above that Bluestein's convolution through HBM; anything else
 * is GOMEL_ERR_UNSUPPORTED_N. */
int gomel_create(const gomel_config* cfg, gomel_handle** out);

/* Same, from a table blob exported by another handle (the multi-GPU path: rank 0 builds, the blob is broadcast
 * over RCCL/xGMI, every rank creates from it).  Fails with GOMEL_ERR_BAD_TABLES if it does not match cfg. */
int gomel_create_from_tables(const gomel_config* cfg, const void* blob, size_t blob_bytes, gomel_handle** out);
size_t gomel_tables_bytes(const gomel_handle* h);
int gomel_tables_export(const gomel_handle* h, void* blob, size_t cap);
/* The same blob straight from a configuration, on the host alone (no device is touched): what rank 0 can build before any
 * GPU work, and what lets the table logic (undomel / domel index math, mel/impl.go:311-322,348-359; gossp's Hann) be checked
 * on a machine without a GPU.  gomel_tables_bytes_for returns 0 for an invalid configuration. */
size_t gomel_tables_bytes_for(const gomel_config* cfg);
int gomel_tables_build(const gomel_config* cfg, void* blob, size_t cap);

/* Inject the analysis/synthesis window (s.Window, mel/mel.go:92,120): n must equal Resolut. */
int gomel_set_window(gomel_handle* h, const double* window, size_t n);
int gomel_get_window(const gomel_handle* h, double* window, size_t cap);

/* Destroy every gomel_batch created from the handle first, and wait for every submitted job: with jobs outstanding the call is
 * refused (GOMEL_ERR_BUSY) and the handle stays valid. */
int gomel_destroy(gomel_handle* h);

/* len(reconstructedSignal), mel/mel.go:80.  0 on bad arguments. */
size_t gomel_out_len(const gomel_config* cfg, size_t n_pairs);

/* (*Mel).FromMel, mel/mel.go:142-152 = spectral_denormalize + undomel + undospectrum + ISTFT (Griffin-Lim).
 *   mel_pairs : [n_pairs][2] log-domain mel, NOT modified (the Go shim applies the reference's in-place exp side
 *               effect, mel/mel.go:143, to the caller's slice itself).
 *   x0        : initial signal, L doubles, replacing rand.Float64() (mel/mel.go:81-83); NULL = generate
 *               x0[i] = (splitmix64(seed + (i+1)*0x9E3779B97F4A7C15) >> 11) * 2^-53.
 *   out       : L doubles; *out_len receives L. */
int gomel_from_mel(gomel_handle* h, const double* mel_pairs, size_t n_pairs, const double* x0, uint64_t seed,
                   double* out, size_t out_cap, size_t* out_len);

/* ISTFT(s, spectrogram, numIterations), mel/mel.go:76-139 -- the Griffin-Lim loop itself, exported by the reference.  Of its
 * complex [F][N] argument it reads only |spectrogram[f][j]| for j <= N/2 (phases are re-estimated every iteration and the upper
 * half is conjugate-filled, mel.go:98-108), so the boundary takes those magnitudes: [n_frames][N/2+1] doubles (host or device).
 * An ODD frame length N keeps one more bin: the fill `for j := 1; j < N/2` (mel.go:105) reaches bins N-1 ... N/2+2 only, so bins
 * N/2 AND N/2+1 keep magnitudes of their own and the rows are N/2+2 long -- gomel_istft_row_len says which.
 * Window and hop come from the handle (gomel_set_window injects s.Window); `iterations` = numIterations >= 0 replaces the
 * handle's GriffinLimIterations for this call.  x0 / seed / out as for gomel_from_mel. */
size_t gomel_istft_row_len(const gomel_config* cfg); /* N/2+1, or N/2+2 when Resolut is odd; 0 for a NULL cfg */
int gomel_istft(gomel_handle* h, const double* magnitudes, size_t n_frames, int32_t iterations, const double* x0, uint64_t seed,
                double* out, size_t out_cap, size_t* out_len);
/* The same with ISTFT's side effect on its argument (mel/mel.go:98-108: every iteration overwrites spectrogram[i][j] with
 * Rect(Abs(spectrogram[i][j]), Phase(S[j])) and conjugate-fills the upper half): spec_out receives [n_frames][row_len][2] doubles
 * (row_len = gomel_istft_row_len: N/2+1, odd N: N/2+2), the bins that keep their own magnitude, of what the reference leaves in the caller's slice -- the last iteration's phase estimate on the caller's
 * magnitudes; the Go shim writes them back and mirrors bins N/2+1 ... N-1 as conjugates.  Nothing is written when iterations == 0
 * (the reference's loop does not run).  spec_out may be NULL (= gomel_istft).  Runs the generic kernels (any batch size). */
int gomel_istft_spec(gomel_handle* h, const double* magnitudes, size_t n_frames, int32_t iterations, const double* x0, uint64_t seed,
                     double* out, size_t out_cap, size_t* out_len, double* spec_out);

/* Many independent FromMel calls in one go (utterances are independent: mel/mel.go:142-152 keeps no state).
 * x0 may be NULL, and so may any x0[u]; seeds may be NULL (then seed = u). */
int gomel_from_mel_batch(gomel_handle* h, size_t n_utt, const double* const* mel_pairs, const size_t* n_pairs,
                         const double* const* x0, const uint64_t* seeds, double* const* out, const size_t* out_cap);

/* fp32 host I/O: the same call for callers whose spectrograms and waveforms are float (an acoustic model's output,
 * a float PCM sink): half the PCIe bytes.  The log-mel values are taken as given (converted to double on the device, then the
 * fp64 table arithmetic of gomel_from_mel); the iterations run in the handle's precision mode (automatic schedule, fp32, fp64: float
 * at the boundary, the mode's arithmetic inside) and the waveform is rounded to float once, at the end. */
int gomel_from_mel_batch_f32(gomel_handle* h, size_t n_utt, const float* const* mel_pairs, const size_t* n_pairs,
                             const float* const* x0, const uint64_t* seeds, float* const* out, const size_t* out_cap);

/* Calls in flight (SURVEY 8(b) "Threading", 8(d) metric: wall time of gomel_from_mel_batch INCLUDING H2D and D2H).  A bulk caller
 * that keeps two calls in flight hides every copy behind the other call's kernels: gomel_from_mel_batch_submit queues one call --
 * upload, every kernel in whole-batch launches, download, on three streams -- and returns; gomel_job_wait blocks until the waveforms
 * are in `out` and frees the job (every submitted job must be waited for, in any order; a wait returns the call's status).
 * `mel_pairs[u]`, `x0[u]` and `out[u]` must stay valid and untouched until the wait returns; the pointer ARRAYS themselves are read
 * during submit only.  Arguments as gomel_from_mel_batch.  A call too small for the fused kernels runs inside submit and comes back
 * as a finished job.  gomel_job_poll: *finished = 1 once a wait would not block.
 * gomel_set_call_mode(h, 1) makes the synchronous gomel_from_mel_batch / _f32 do the same (one submit + wait per call): for hosts
 * that overlap calls on one handle from several threads (two goroutines calling in a loop then pipeline at whole-batch granularity);
 * 0 (default) = the chunked call, the best a lone synchronous caller gets.  The two modes cut utterances into different segments,
 * so their results differ at rounding level (other summation order of the overlap-add); within a mode results are bit-reproducible. */
typedef struct gomel_job gomel_job;
int gomel_from_mel_batch_submit(gomel_handle* h, size_t n_utt, const double* const* mel_pairs, const size_t* n_pairs,
                                const double* const* x0, const uint64_t* seeds, double* const* out, const size_t* out_cap, gomel_job** job);
int gomel_from_mel_batch_submit_f32(gomel_handle* h, size_t n_utt, const float* const* mel_pairs, const size_t* n_pairs,
                                    const float* const* x0, const uint64_t* seeds, float* const* out, const size_t* out_cap, gomel_job** job);
int gomel_job_wait(gomel_job* job);
/* The bounded wait a server uses (a goroutine must not block forever behind a device that has stopped answering): as gomel_job_wait,
 * but gives up after `timeout_ms` milliseconds with GOMEL_ERR_BUSY -- the job is then STILL VALID and still owns the caller's buffers
 * (wait again, poll, or give the process up; a job cannot be cancelled: its copies may be in flight).  Any other status: the job is
 * finished and freed exactly as by gomel_job_wait.  timeout_ms <= 0 = one poll. */
int gomel_job_wait_timeout(gomel_job* job, int64_t timeout_ms);
int gomel_job_poll(gomel_job* job, int* finished);
int gomel_set_call_mode(gomel_handle* h, int mode);

/* One call, several GPUs of one node (SURVEY 8e "one host thread/stream per device"): `handles` holds one handle per device,
 * all created from the same configuration (gomel_create_from_tables gives them bit-identical tables); the utterances are cut
 * into contiguous shards balanced by frame count, each shard runs gomel_from_mel_batch on its handle from its own host thread.
 * No collective on the data path; default seeds are the GLOBAL utterance indices, so results do not depend on the sharding.
 * Each shard's host thread first binds itself to the CPUs of the NUMA node its device hangs off (gomel_numa_bind_thread with the
 * device's PCI bus id; GOMEL_NUMA_BIND=0 switches that off), so that its pinning and copy submission run beside the GPU's root port. */
int gomel_from_mel_batch_multi(gomel_handle* const* handles, size_t n_handles, size_t n_utt, const double* const* mel_pairs,
                               const size_t* n_pairs, const double* const* x0, const uint64_t* seeds, double* const* out,
                               const size_t* out_cap);

/* Pinned host memory for bulk callers.  gomel_from_mel_batch and friends pin pageable caller buffers for the duration of a
 * call (hipHostRegister: ~8 ms per GiB, every call); a caller that reuses its buffers allocates them here, or registers its
 * own once, and pays nothing per call.  gomel_host_unregister takes the pointer that was registered (a pointer it does not know:
 * GOMEL_ERR_BAD_CONFIG); a range inside an already pinned one shares that registration and releases its share of it. */
void* gomel_host_alloc(size_t bytes);
int gomel_host_free(void* p);
int gomel_host_register(void* p, size_t bytes);
int gomel_host_unregister(void* p);

/* How gomel_from_mel_batch / _f32 will cut a call with these utterance lengths into the chunks it pipelines (upload c+1 | compute
 * c | download c-1): writes *n_chunks + 1 utterance indices 0 = bounds[0] < ... < bounds[*n_chunks] = n_utt.  Host logic only (no
 * device needed); bounds needs room for at most 66 entries (cap = its capacity).  The plan: half-size first and last chunks, full
 * chunks of ~12k frames at Resolut 4096 / ~32k below -- twice that for calls that run fp64 iterations (their kernels take longer per
 * frame, so larger chunks still hide the copies) --, boundaries moved by up to two utterances to where a chunk fills the fused
 * kernel's rounds; GOMEL_CHUNKS / GOMEL_CHUNK_SHARES override it.  gomel_plan_chunks: a handle in its default (automatic) precision
 * mode; gomel_plan_chunks_mode: a call with that many fp64 iterations (0 = the fp32 fast mode, GOMEL_SCHEDULE_AUTO = the default). */
int gomel_plan_chunks(const gomel_config* cfg, size_t n_utt, const size_t* n_pairs, size_t* bounds, size_t cap, size_t* n_chunks);
int gomel_plan_chunks_mode(const gomel_config* cfg, int32_t fp64_iterations, size_t n_utt, const size_t* n_pairs, size_t* bounds, size_t cap, size_t* n_chunks);

/* The same work split so that inputs can be resident in HBM before the clock starts (bench / pipelines):
 * create (plan + allocate) -> upload (H2D of log-mel and x0) -> run (all device work, asynchronous on the
 * handle's stream) -> download (D2H of the waveforms, synchronises). */
int gomel_batch_create(gomel_handle* h, size_t n_utt, const size_t* n_pairs, gomel_batch** out);
int gomel_batch_create_f32(gomel_handle* h, size_t n_utt, const size_t* n_pairs, gomel_batch** out); /* float log-mel / x0 / waveforms */
int gomel_batch_upload(gomel_batch* b, const double* const* mel_pairs, const double* const* x0, const uint64_t* seeds);
int gomel_batch_upload_f32(gomel_batch* b, const float* const* mel_pairs, const float* const* x0, const uint64_t* seeds);
int gomel_batch_run(gomel_batch* b);
int gomel_batch_run_timed(gomel_batch* b, gomel_timing* t); /* synchronises */
int gomel_batch_sync(gomel_batch* b);
int gomel_batch_download(gomel_batch* b, double* const* out, const size_t* out_cap);
int gomel_batch_download_f32(gomel_batch* b, float* const* out, const size_t* out_cap);
int gomel_batch_destroy(gomel_batch* b);
/* 0 = let the library choose (fused kernel when the hop allows it and the batch is big enough to fill the chip: from 1.5 k
 * frames at Resolut 4096 to 12 k at 256, measured crossovers);
 * persistent kernel -- every iteration in one launch -- when all frames of the batch fit the chip at once, i.e. one utterance
 * or a few short ones, the hop is a multiple of 4 and at least Resolut/16; generic kernels otherwise);
 * 1 = force the generic kernels; 2 = require a fused kernel (k_gl_fused when the hop is SH*Resolut/16, SH in 1,2,4,5,8; its any-hop
 * form for every other hop from Resolut/32 up to Resolut - 1; error for smaller hops and hops >= Resolut);
 * 3 = require the persistent kernel (gomel_batch_create fails with GOMEL_ERR_BAD_CONFIG for a batch that does not fit it).
 * The persistent kernel's workgroups wait for each other inside the launch: should another context keep them from all being
 * resident, the launch gives up after 20 ms, the batch is recomputed with the generic kernels (same results, bit for bit) and
 * the handle leaves that path alone for its next 64 batches (doubling with every further timeout, up to 4096). */
int gomel_set_path(gomel_handle* h, int mode);
/* Arithmetic of the Griffin-Lim iterations (mel/mel.go:85-136 is float64 / complex128 throughout) for batches created afterwards:
 *   0  (default) AUTOMATIC: the calibrated precision schedule gomel_auto_fp64_iterations(cfg, I).  Contract: calibrated, not
 *      guaranteed -- the call ends within 1e-4 RMS of the reference's float64 arithmetic except on about one utterance in 10^5
 *      (random spectrograms, 60 iterations, 2.1 M held out: 1.5e-6 at the towav geometry, 2e-5 at the TTS geometry; Griffin-Lim
 *      amplifies rounding without bound on a few inputs; measurements: DESIGN.md section 2);
 *   32 fp32 kernels for every iteration: the opt-in FAST mode.  Typically 1e-6 from the
 *      fp64 arithmetic, but ~1 % of 60-iteration utterances end beyond 1e-4;
 *   64 fp64 kernels for every iteration (~1e-13 from the fp64 restatement of the reference): the mode that is guaranteed on every input. */
int gomel_set_precision(gomel_handle* h, int bits);
/* Precision schedule: the first `fp64_iterations` Griffin-Lim iterations of every call run the fp64 kernels, the remaining ones the
 * fp32 kernels (the signal is rounded to fp32 once, at the hand-over).  Griffin-Lim on a spectrogram that is not consistent
 * amplifies a rounding-level perturbation by every iteration that FOLLOWS it, and it is most unstable in its first iterations, from
 * the random initial signal (mel/mel.go:80-83): fp32 rounding confined to the later iterations stays near 1e-7 where fp32
 * throughout can end 1e-4 away from the reference's float64 arithmetic.
 *   GOMEL_SCHEDULE_AUTO = the calibrated schedule (= gomel_set_precision(h, 0), the default);
 *   0 = fp32 throughout (= gomel_set_precision(h, 32)); negative, or not below the call's iteration count = fp64 throughout (= 64).
 * A schedule runs on every kernel path of the register FFT cores (fused, persistent, generic pair: a 64-bit call of k iterations
 * followed by an fp32 call started from its output, bit for bit); the generic frame lengths (their kernels are fp64) and
 * gomel_istft_spec run fp64 throughout whenever the schedule asks for any fp64 iteration at all. */
#define GOMEL_SCHEDULE_AUTO (-2147483647 - 1)
int gomel_set_precision_schedule(gomel_handle* h, int32_t fp64_iterations);
/* The automatic schedule: of a call of `iterations` Griffin-Lim iterations, how many run in fp64 first.  Host logic only.
 * Two calibrated tables over the iteration count (interpolated upwards between calibrated counts; the trailing fp32 iterations are
 * held beyond the last one; up to five iterations -- the reference's shipped 2, mel/mel.go:39, among them -- run in fp64 throughout):
 * one for frames that overlap up to four-fold (Resolut <= 4 * Window), one from six-fold on (Resolut >= 6 * Window, NewMel's
 * defaults, mel/mel.go:30-41); in between the result is interpolated linearly in the overlap Resolut / Window and rounded up, so the
 * rule is monotone in the iteration count and in the overlap.  The tables and the data behind them: gomel_hip.hip, DESIGN.md section 2.
 * Generic frame lengths: = iterations (their kernels are fp64).  Negative iterations / NULL cfg: 0. */
int32_t gomel_auto_fp64_iterations(const gomel_config* cfg, int32_t iterations);
/* What a call of `iterations` iterations (negative: the handle's GriffinLimIterations) on this handle will do under its current
 * mode: *fp64_iterations = 0 (fp32 throughout) ... iterations (fp64 throughout). */
int gomel_get_precision(const gomel_handle* h, int32_t iterations, int32_t* fp64_iterations);

/* (*Mel).ToMel, mel/mel.go:46-74 = pad + gossp STFT + |S[j]|,|S[N-1-j]| + domel + ln(max(v,1e-5)). */
size_t gomel_to_mel_pairs(const gomel_config* cfg, size_t n_samples);
int gomel_to_mel(gomel_handle* h, const double* wave, size_t n_samples, double* out_pairs, size_t cap_pairs,
                 size_t* n_pairs);
/* Many ToMel calls in one go (independent waveforms, mel/mel.go:46-74 keeps no state): one STFT launch and one domel launch over
 * all frames -- what generates FromMel inputs at scale.  n_pairs_out[u] (may be NULL) receives len(ToMel(wave[u])). */
int gomel_to_mel_batch(gomel_handle* h, size_t n_utt, const double* const* wave, const size_t* n_samples, double* const* out_pairs,
                       const size_t* cap_pairs, size_t* n_pairs_out);

/* Package phase (the reference's phase-preserving codec, phase/phase.go) on the same handle: the handle supplies Window, Resolut
 * and the Hann table; NumMels / MelFmax are not used.  NumFreqs and VolumeBoost are the fields of `type Phase` (phase.go:8-18).
 *   gomel_to_phase   = (*Phase).ToPhase, phase/phase.go:41-72: pad + STFT + (Im S[j+1], Re S[N-1-j]) + shrink to NumFreqs.
 *   gomel_from_phase = (*Phase).FromPhase, phase/phase.go:137-155: grow + undospectrum + ISTFT with window-sum
 *                      normalisation (phase.go:96-134) + VolumeBoost.  fp64 throughout. */
size_t gomel_phase_pairs(const gomel_config* cfg, int32_t num_freqs, size_t n_samples);
size_t gomel_phase_out_len(const gomel_config* cfg, int32_t num_freqs, size_t n_pairs);
int gomel_to_phase(gomel_handle* h, int32_t num_freqs, const double* wave, size_t n_samples, double* out_pairs, size_t cap_pairs,
                   size_t* n_pairs);
int gomel_from_phase(gomel_handle* h, int32_t num_freqs, double volume_boost, const double* pairs, size_t n_pairs, double* out,
                     size_t out_cap, size_t* out_len);

/* pad / isPadded, mel/impl.go:429-479 (ToWavPng's trim decision, mel/mel.go:228). */
size_t gomel_pad_len(size_t n, int32_t window);
int gomel_is_padded(size_t original_len, size_t padded_len, int32_t window);

/* Test hooks for the FFT core (what fft.FFTReal / fft.IFFT do per frame, mel/mel.go:95,116):
 * frames [n_frames][N] float -> spectrum [n_frames][N/2+1][2] float (unnormalised forward), and back (1/N). */
int gomel_rfft_frames(gomel_handle* h, const float* frames, size_t n_frames, float* spectrum);
int gomel_irfft_frames(gomel_handle* h, const float* spectrum, size_t n_frames, float* frames);
/* The magnitudes the iteration clamps to: [F][N/2+1] float, already divided by N (the iFFT scale is folded in). */
int gomel_target_magnitudes(gomel_handle* h, const double* mel_pairs, size_t n_pairs, float* mag_out);

const char* gomel_strerror(int status);
int gomel_last_hip_error(void);
int gomel_abi_version(void);

/* NUMA locality of host threads (host-side PCIe traffic, not xGMI, limits the 8-GPU case).  Host logic only.
 *   gomel_numa_cpus: the NUMA node of the PCI device `pci_bus_id` ("0000:85:00.0", from hipDeviceGetPCIBusId) and that node's CPUs,
 *     read from <sysfs_root>/bus/pci/devices/<bdf>/numa_node and <sysfs_root>/devices/system/node/node<N>/cpulist (sysfs_root NULL
 *     = "/sys"; injectable for tests).  *n_cpus receives the count even when cap is smaller.  GOMEL_ERR_EMPTY when the machine does not
 *     say (no such device, numa_node = -1, no cpulist): not an error for the caller, there is simply nothing to bind to.
 *   gomel_numa_bind_thread: restricts the CALLING thread to those CPUs, intersected with its current affinity mask (never widens
 *     it); *n_bound = CPUs it now runs on.  GOMEL_ERR_EMPTY as above, or when the intersection is empty (the thread is left alone). */
int gomel_numa_cpus(const char* sysfs_root, const char* pci_bus_id, int32_t* node, int32_t* cpus, size_t cap, size_t* n_cpus);
int gomel_numa_bind_thread(const char* sysfs_root, const char* pci_bus_id, int32_t* node, int32_t* n_bound);
int gomel_device_count(void);
int gomel_device_name(const gomel_handle* h, char* buf, size_t cap);
/* Blocks until every piece of work queued on `device` by this process has finished (hipDeviceSynchronize with the calling thread's
 * device saved and restored): what a timing harness or a host without its own HIP binding brackets a measurement with.
 * device = -1: the calling thread's current device. */
int gomel_device_synchronize(int32_t device);

#ifdef __cplusplus
}
#endif
#endif /* GOMEL_HIP_H */
