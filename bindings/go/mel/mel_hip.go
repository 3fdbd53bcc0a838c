// Package mel: drop-in replacement for the hot path of github.com/neurlang/gomel/mel, calling libgomel_hip.so
// (include/gomel_hip.h) through cgo.  Same exported names and field set as the reference's `type Mel`
// (mel/mel.go:10-27); ToMel / FromMel / ISTFT run on the GPU, everything else (PNG/WAV codec, pad, ToWavPng's glue) stays
// the reference's own Go code and is not repeated here.
//
// `Mel` stays what it is in the reference: a plain value of exported fields that callers may copy (`m2 := *m`), compare and
// drop without ceremony.  Device handles therefore do NOT live in the struct: they sit in a small package-level cache keyed by
// the configuration (at most maxEngines live handles, least recently used evicted once idle), so there is no mutex to copy
// (go vet copylocks), no double free between copies and nothing to Close.  ReleaseEngines() frees the cache explicitly.
//
// NOT COMPILED IN THIS REPOSITORY'S PIPELINE: there is no Go toolchain in the build image (SURVEY Appendix E).  The file
// is the binding a maintainer adds next to mel/mel.go, replacing the bodies of ToMel, FromMel and ISTFT (INTEGRATION.md).
// Build: CGO_CFLAGS=-I<repo>/include CGO_LDFLAGS="-L<repo>/gomel_amd -lgomel_hip -Wl,-rpath,<repo>/gomel_amd".
package mel

/*
#cgo LDFLAGS: -lgomel_hip
#include <stdlib.h>
#include "gomel_hip.h"
*/
import "C"

import (
	"errors"
	"math"
	"math/cmplx"
	"math/rand"
	"runtime"
	"sync"
	"time"
	"unsafe"

	"github.com/r9y9/gossp/stft"
)

// Mel represents the configuration for generating mel spectrograms (field for field the reference's struct).
type Mel struct {
	NumMels  int
	MelFmin  float64
	MelFmax  float64
	TuneMul  float64
	TuneAdd  float64
	Window   int
	Resolut  int
	YReverse bool

	GriffinLimIterations int
	VolumeBoost          float64
	SampleRate           int
}

// NewMel creates a new Mel instance with default values (mel/mel.go:30-41).
func NewMel() *Mel {
	return &Mel{NumMels: 160, MelFmin: 0, MelFmax: 8000, TuneMul: 1, TuneAdd: 0, Window: 256, Resolut: 2048, GriffinLimIterations: 2}
}

var ErrFileNotLoaded = errors.New("wavNotLoaded")

func statusErr(rc C.int) error {
	if rc == C.GOMEL_OK {
		return nil
	}
	if rc == C.GOMEL_ERR_BUSY {
		return ErrBusy
	}
	return errors.New("gomel_hip: " + C.GoString(C.gomel_strerror(rc)))
}

// ErrBusy: gomel_destroy was called on a handle with submitted jobs that have not been waited for (GOMEL_ERR_BUSY); the handle stays valid.
var ErrBusy = errors.New("gomel_hip: handle busy: submitted jobs have not been waited for")

// ---- handle cache -------------------------------------------------------------------------------------------------------

// Fp64Iterations selects the arithmetic of the Griffin-Lim iterations for the calls that follow (not in the reference, whose
// arithmetic is float64 throughout; the struct Mel stays field for field the reference's, so this is a package variable):
// Fp64Auto (the default) = the library's calibrated schedule, C.gomel_auto_fp64_iterations: the first k(GriffinLimIterations)
// iterations in fp64, the rest in fp32 -- within 1e-4 RMS of the float64 result on every utterance of the calibration sets (35 of 60,
// 50 of 60 where frames overlap six-fold or more as in NewMel's defaults, interpolated in the overlap between four- and six-fold; both of the reference's shipped 2); 0 = the fp32 kernels throughout (the fast mode; on spectrograms that are not consistent ~1 % of
// 60-iteration utterances end more than 1e-4 RMS from the float64 result), -1 = the fp64 kernels throughout (agrees with float64
// arithmetic on every input), k > 0 = the first k iterations in fp64, the rest in fp32 (C.gomel_set_precision_schedule).
const Fp64Auto = math.MinInt32 // C.GOMEL_SCHEDULE_AUTO

var Fp64Iterations = Fp64Auto

// WholeBatchCalls = true makes every FromMelBatch call one whole batch on the device (C.gomel_set_call_mode(h, 1)) instead of a
// chunked one: for programs that call FromMelBatch from two or more goroutines on the same configuration, whose calls then
// pipeline -- the copies of one call beside the kernels of the other (INTEGRATION.md).  A lone caller leaves it false.
var WholeBatchCalls = false

type engineKey struct {
	numMels, window, resolut, iters, fp64 int32
	whole                                 bool
	fmin, fmax, tuneMul, tuneAdd         float64
}

type engine struct {
	h    *C.gomel_handle
	cfg  C.gomel_config
	refs int    // calls in flight (the C library serialises them on its own mutex)
	used uint64 // LRU stamp
}

const maxEngines = 8

// JobTimeout bounds FromMelBatch's wait for the device when WholeBatchCalls is set (gomel_job_wait_timeout); a call still running then returns ErrBusy.
var JobTimeout = 10 * time.Minute

var (
	enginesMu sync.Mutex
	engines   = map[engineKey]*engine{}
	useClock  uint64
)

func (m *Mel) key() engineKey {
	return engineKey{int32(m.NumMels), int32(m.Window), int32(m.Resolut), int32(m.GriffinLimIterations), int32(Fp64Iterations), WholeBatchCalls, m.MelFmin, m.MelFmax, m.TuneMul, m.TuneAdd}
}

// acquire returns the handle for m's current field values, creating it on first use; release() must follow.
func (m *Mel) acquire() (*engine, error) {
	k := m.key()
	enginesMu.Lock()
	defer enginesMu.Unlock()
	useClock++
	if e, ok := engines[k]; ok {
		e.refs++
		e.used = useClock
		return e, nil
	}
	var c C.gomel_config
	C.gomel_default_config(&c)
	c.num_mels, c.window, c.resolut = C.int32_t(m.NumMels), C.int32_t(m.Window), C.int32_t(m.Resolut)
	c.griffin_lim_iterations = C.int32_t(m.GriffinLimIterations)
	c.mel_fmin, c.mel_fmax, c.tune_mul, c.tune_add = C.double(m.MelFmin), C.double(m.MelFmax), C.double(m.TuneMul), C.double(m.TuneAdd)
	e := &engine{cfg: c, refs: 1, used: useClock}
	if rc := C.gomel_create(&c, &e.h); rc != C.GOMEL_OK {
		return nil, statusErr(rc)
	}
	if rc := C.gomel_set_precision_schedule(e.h, C.int32_t(k.fp64)); rc != C.GOMEL_OK {
		C.gomel_destroy(e.h)
		return nil, statusErr(rc)
	}
	if k.whole {
		if rc := C.gomel_set_call_mode(e.h, 1); rc != C.GOMEL_OK {
			C.gomel_destroy(e.h)
			return nil, statusErr(rc)
		}
	}
	engines[k] = e
	for len(engines) > maxEngines { // evict idle handles, least recently used first
		var victimK engineKey
		var victim *engine
		for kk, ee := range engines {
			if ee.refs == 0 && (victim == nil || ee.used < victim.used) {
				victimK, victim = kk, ee
			}
		}
		if victim == nil {
			break
		}
		if destroy(victim.h) != nil {
			break // busy: stays in the cache, a later call evicts it
		}
		delete(engines, victimK)
	}
	return e, nil
}

// destroy frees a handle.  gomel_destroy refuses (GOMEL_ERR_BUSY) while submitted jobs have not been waited for -- the handle then
// stays valid and must be kept: dropping it would leak it together with its device buffers.
func destroy(h *C.gomel_handle) error {
	return statusErr(C.gomel_destroy(h))
}

func (e *engine) release() {
	enginesMu.Lock()
	e.refs--
	enginesMu.Unlock()
}

// ReleaseEngines destroys every idle device handle of the cache (no counterpart in the reference, which holds no resources).
func ReleaseEngines() {
	enginesMu.Lock()
	defer enginesMu.Unlock()
	for k, e := range engines {
		if e.refs == 0 && destroy(e.h) == nil {
			delete(engines, k)
		}
	}
}

// ---- the hot path -------------------------------------------------------------------------------------------------------

// ToMel generates a mel spectrogram from a wave buffer (mel/mel.go:46-74).
func (m *Mel) ToMel(buf []float64) ([][2]float64, error) {
	if len(buf) == 0 {
		return nil, statusErr(C.GOMEL_ERR_EMPTY)
	}
	e, err := m.acquire()
	if err != nil {
		return nil, err
	}
	defer e.release()
	n := int(C.gomel_to_mel_pairs(&e.cfg, C.size_t(len(buf))))
	out := make([][2]float64, n)
	if n == 0 {
		return out, statusErr(C.GOMEL_ERR_TOO_SHORT)
	}
	var got C.size_t
	rc := C.gomel_to_mel(e.h, (*C.double)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)),
		(*C.double)(unsafe.Pointer(&out[0][0])), C.size_t(n), &got)
	runtime.KeepAlive(buf)
	runtime.KeepAlive(out)
	return out[:int(got)], statusErr(rc)
}

// FromMel generates a wave buffer from a mel spectrogram (mel/mel.go:142-152).  A [][2]float64 is one contiguous
// block of 2*len float64, passed without copying; the initial signal comes from math/rand's global source exactly as
// in the reference (mel/mel.go:81-83), so seeding behaves as before.
func (m *Mel) FromMel(ospectrum [][2]float64) ([]float64, error) {
	if len(ospectrum) == 0 {
		return nil, statusErr(C.GOMEL_ERR_EMPTY) // the reference panics here (mel/mel.go:78)
	}
	e, err := m.acquire()
	if err != nil {
		return nil, err
	}
	defer e.release()
	L := int(C.gomel_out_len(&e.cfg, C.size_t(len(ospectrum))))
	if L == 0 {
		return nil, statusErr(C.GOMEL_ERR_BAD_LENGTH) // the reference panics here (mel/impl.go:366)
	}
	x0 := make([]float64, L)
	for i := range x0 {
		x0[i] = rand.Float64()
	}
	out := make([]float64, L)
	var got C.size_t
	rc := C.gomel_from_mel(e.h, (*C.double)(unsafe.Pointer(&ospectrum[0][0])), C.size_t(len(ospectrum)),
		(*C.double)(unsafe.Pointer(&x0[0])), 0, (*C.double)(unsafe.Pointer(&out[0])), C.size_t(L), &got)
	runtime.KeepAlive(ospectrum)
	runtime.KeepAlive(x0)
	runtime.KeepAlive(out)
	// side effect of spectral_denormalize on the caller's slice (mel/mel.go:143, mel/impl.go:421-427)
	for i := range ospectrum {
		ospectrum[i][0] = math.Exp(ospectrum[i][0])
		ospectrum[i][1] = math.Exp(ospectrum[i][1])
	}
	return out[:int(got)], statusErr(rc)
}

// ISTFT is the Griffin-Lim loop itself (mel/mel.go:76-139).  Of `spectrogram` the reference reads only the magnitudes of
// bins 0..frameLen/2 (phases are re-estimated, the upper half is conjugate-filled), so those are what crosses the boundary;
// s.Window is injected as the analysis/synthesis window.  As in the reference (mel/mel.go:98-108) the caller's spectrogram
// holds the last iteration's phase estimate afterwards (gomel_istft_spec returns bins 0..frameLen/2, the rest is mirrored here).
func ISTFT(s *stft.STFT, spectrogram [][]complex128, numIterations int) []float64 {
	numFrames, frameLen := len(spectrogram), len(spectrogram[0]) // panics on an empty spectrogram, as the reference does
	var c C.gomel_config
	C.gomel_default_config(&c)
	c.resolut, c.window = C.int32_t(frameLen), C.int32_t(s.FrameShift)
	var h *C.gomel_handle
	if rc := C.gomel_create(&c, &h); rc != C.GOMEL_OK {
		panic(statusErr(rc))
	}
	defer C.gomel_destroy(h)
	if rc := C.gomel_set_window(h, (*C.double)(unsafe.Pointer(&s.Window[0])), C.size_t(len(s.Window))); rc != C.GOMEL_OK {
		panic(statusErr(rc))
	}
	half := frameLen/2 + 1 + frameLen%2 // C.gomel_istft_row_len: an odd frameLen keeps bins N/2 and N/2+1 (mel.go:105 fills N-1 ... N/2+2 only)
	mags := make([]float64, numFrames*half)
	for i := range spectrogram {
		for j := 0; j < half; j++ {
			mags[i*half+j] = cmplx.Abs(spectrogram[i][j])
		}
	}
	L := frameLen + (numFrames-1)*s.FrameShift
	x0 := make([]float64, L)
	for i := range x0 {
		x0[i] = rand.Float64()
	}
	out := make([]float64, L)
	spec := make([]float64, numFrames*half*2)
	var got C.size_t
	rc := C.gomel_istft_spec(h, (*C.double)(unsafe.Pointer(&mags[0])), C.size_t(numFrames), C.int32_t(numIterations),
		(*C.double)(unsafe.Pointer(&x0[0])), 0, (*C.double)(unsafe.Pointer(&out[0])), C.size_t(L), &got,
		(*C.double)(unsafe.Pointer(&spec[0])))
	runtime.KeepAlive(mags)
	runtime.KeepAlive(x0)
	runtime.KeepAlive(out)
	runtime.KeepAlive(spec)
	if rc != C.GOMEL_OK {
		panic(statusErr(rc))
	}
	if numIterations > 0 { // the reference's loop body is what writes the spectrogram
		for i := range spectrogram {
			for j := 0; j < half; j++ {
				spectrogram[i][j] = complex(spec[(i*half+j)*2], spec[(i*half+j)*2+1])
			}
			for j := 1; j < frameLen/2; j++ {
				spectrogram[i][frameLen-j] = cmplx.Conj(spectrogram[i][j]) // mel/mel.go:105-108
			}
		}
	}
	return out[:int(got)]
}

// FromMelBatch runs many independent FromMel calls in one go (no counterpart in the reference; the throughput mode).  With
// several handles -- one per GPU, see INTEGRATION.md -- C.gomel_from_mel_batch_multi shards the utterances over the devices.
func (m *Mel) FromMelBatch(spectra [][][2]float64) ([][]float64, error) {
	n := len(spectra)
	if n == 0 {
		return nil, nil
	}
	e, err := m.acquire()
	if err != nil {
		return nil, err
	}
	defer e.release()
	// pointer arrays handed to C must not be Go memory holding Go pointers: build them in C memory, pin the slices
	ptrBytes := C.size_t(unsafe.Sizeof(uintptr(0)))
	mels := (*[1 << 28]*C.double)(C.malloc(C.size_t(n) * ptrBytes))
	x0s := (*[1 << 28]*C.double)(C.malloc(C.size_t(n) * ptrBytes))
	outs := (*[1 << 28]*C.double)(C.malloc(C.size_t(n) * ptrBytes))
	lens := (*[1 << 28]C.size_t)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(C.size_t(0)))))
	defer C.free(unsafe.Pointer(mels))
	defer C.free(unsafe.Pointer(x0s))
	defer C.free(unsafe.Pointer(outs))
	defer C.free(unsafe.Pointer(lens))
	var pin runtime.Pinner
	defer pin.Unpin()
	res := make([][]float64, n)
	init := make([][]float64, n)
	for u, sp := range spectra {
		if len(sp) == 0 {
			return nil, statusErr(C.GOMEL_ERR_EMPTY)
		}
		L := int(C.gomel_out_len(&e.cfg, C.size_t(len(sp))))
		if L == 0 {
			return nil, statusErr(C.GOMEL_ERR_BAD_LENGTH)
		}
		res[u] = make([]float64, L)
		init[u] = make([]float64, L)
		for i := range init[u] {
			init[u][i] = rand.Float64() // the stream the reference would have drawn, utterance after utterance
		}
		pin.Pin(&sp[0][0])
		pin.Pin(&res[u][0])
		pin.Pin(&init[u][0])
		mels[u] = (*C.double)(unsafe.Pointer(&sp[0][0]))
		x0s[u] = (*C.double)(unsafe.Pointer(&init[u][0]))
		outs[u] = (*C.double)(unsafe.Pointer(&res[u][0]))
		lens[u] = C.size_t(len(sp))
	}
	var rc C.int
	if !WholeBatchCalls {
		// a lone caller: the chunked synchronous call (its copies hidden behind its own kernels: the best one call at a time gets)
		rc = C.gomel_from_mel_batch(e.h, C.size_t(n), &mels[0], &lens[0], &x0s[0], nil, &outs[0], nil)
	} else {
		// pipelined callers (the handle is in whole-batch call mode): submit + a BOUNDED wait -- a goroutine never blocks without end behind a
		// device that has stopped answering.  A job that outlives JobTimeout is reported as ErrBusy and waited for in the background (it
		// still owns the pinned slices: they stay pinned until it ends).  Same launches, same bits as the synchronous call in this mode.
		var job *C.gomel_job
		rc = C.gomel_from_mel_batch_submit(e.h, C.size_t(n), &mels[0], &lens[0], &x0s[0], nil, &outs[0], nil, &job)
		if rc == C.GOMEL_OK {
			rc = C.gomel_job_wait_timeout(job, C.int64_t(JobTimeout/time.Millisecond))
			if rc == C.GOMEL_ERR_BUSY {
				held := pin
				pin = runtime.Pinner{}
				enginesMu.Lock()
				e.refs++ // the background wait's own reference (the deferred release() balances this call's)
				enginesMu.Unlock()
				go func() {
					C.gomel_job_wait(job)
					held.Unpin()
					e.release()
				}()
				return nil, ErrBusy
			}
		}
	}
	for _, sp := range spectra {
		for i := range sp {
			sp[i][0] = math.Exp(sp[i][0])
			sp[i][1] = math.Exp(sp[i][1])
		}
	}
	return res, statusErr(rc)
}
