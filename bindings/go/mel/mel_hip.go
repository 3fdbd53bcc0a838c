// Package mel: drop-in replacement for the hot path of github.com/neurlang/gomel/mel, calling libgomel_hip.so
// (include/gomel_hip.h) through cgo.  Same exported names and field set as the reference's `type Mel`
// (mel/mel.go:10-27); ToMel / FromMel run on the GPU, everything else (PNG/WAV codec, pad, ToWavPng's glue) stays the
// reference's own Go code and is not repeated here.
//
// NOT COMPILED IN THIS REPOSITORY'S PIPELINE: there is no Go toolchain in the build image (SURVEY Appendix E).  The file
// is the binding a maintainer adds next to mel/mel.go, replacing the bodies of ToMel and FromMel; it is kept small and
// mechanical on purpose.  Build: CGO_CFLAGS=-I<repo>/include CGO_LDFLAGS="-L<repo>/gomel_amd -lgomel_hip".
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
	"sync"
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

	mu     sync.Mutex
	handle *C.gomel_handle
	key    C.gomel_config
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
	return errors.New("gomel_hip: " + C.GoString(C.gomel_strerror(rc)))
}

// engine (re)creates the device handle when the exported fields changed since the last call.
func (m *Mel) engine() (*C.gomel_handle, error) {
	var c C.gomel_config
	C.gomel_default_config(&c)
	c.num_mels, c.window, c.resolut = C.int32_t(m.NumMels), C.int32_t(m.Window), C.int32_t(m.Resolut)
	c.griffin_lim_iterations = C.int32_t(m.GriffinLimIterations)
	c.mel_fmin, c.mel_fmax, c.tune_mul, c.tune_add = C.double(m.MelFmin), C.double(m.MelFmax), C.double(m.TuneMul), C.double(m.TuneAdd)
	if m.handle != nil && c == m.key {
		return m.handle, nil
	}
	if m.handle != nil {
		C.gomel_destroy(m.handle)
		m.handle = nil
	}
	if rc := C.gomel_create(&c, &m.handle); rc != C.GOMEL_OK {
		return nil, statusErr(rc)
	}
	m.key = c
	return m.handle, nil
}

// ToMel generates a mel spectrogram from a wave buffer (mel/mel.go:46-74).
func (m *Mel) ToMel(buf []float64) ([][2]float64, error) {
	m.mu.Lock()
	defer m.mu.Unlock()
	h, err := m.engine()
	if err != nil {
		return nil, err
	}
	if len(buf) == 0 {
		return nil, statusErr(C.GOMEL_ERR_EMPTY)
	}
	n := int(C.gomel_to_mel_pairs(&m.key, C.size_t(len(buf))))
	out := make([][2]float64, n)
	if n == 0 {
		return out, statusErr(C.GOMEL_ERR_TOO_SHORT)
	}
	var got C.size_t
	rc := C.gomel_to_mel(h, (*C.double)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)),
		(*C.double)(unsafe.Pointer(&out[0][0])), C.size_t(n), &got)
	return out[:int(got)], statusErr(rc)
}

// FromMel generates a wave buffer from a mel spectrogram (mel/mel.go:142-152).  A [][2]float64 is one contiguous
// block of 2*len float64, passed without copying; the initial signal comes from math/rand's global source exactly as
// in the reference (mel/mel.go:81-83), so seeding behaves as before.
func (m *Mel) FromMel(ospectrum [][2]float64) ([]float64, error) {
	m.mu.Lock()
	defer m.mu.Unlock()
	h, err := m.engine()
	if err != nil {
		return nil, err
	}
	if len(ospectrum) == 0 {
		return nil, statusErr(C.GOMEL_ERR_EMPTY) // the reference panics here (mel/mel.go:78)
	}
	L := int(C.gomel_out_len(&m.key, C.size_t(len(ospectrum))))
	if L == 0 {
		return nil, statusErr(C.GOMEL_ERR_BAD_LENGTH) // the reference panics here (mel/impl.go:366)
	}
	x0 := make([]float64, L)
	for i := range x0 {
		x0[i] = rand.Float64()
	}
	out := make([]float64, L)
	var got C.size_t
	rc := C.gomel_from_mel(h, (*C.double)(unsafe.Pointer(&ospectrum[0][0])), C.size_t(len(ospectrum)),
		(*C.double)(unsafe.Pointer(&x0[0])), 0, (*C.double)(unsafe.Pointer(&out[0])), C.size_t(L), &got)
	// side effect of spectral_denormalize on the caller's slice (mel/mel.go:143, mel/impl.go:421-427)
	for i := range ospectrum {
		ospectrum[i][0] = math.Exp(ospectrum[i][0])
		ospectrum[i][1] = math.Exp(ospectrum[i][1])
	}
	return out[:int(got)], statusErr(rc)
}

// ISTFT is the Griffin-Lim loop itself (mel/mel.go:76-139).  Of `spectrogram` the reference reads only the magnitudes of
// bins 0..frameLen/2 (phases are re-estimated, the upper half is conjugate-filled), so those are what crosses the boundary;
// s.Window is injected as the analysis/synthesis window.  The reference also leaves its last phase estimate in the caller's
// spectrogram; this shim does not.
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
	half := frameLen/2 + 1
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
	var got C.size_t
	if rc := C.gomel_istft(h, (*C.double)(unsafe.Pointer(&mags[0])), C.size_t(numFrames), C.int32_t(numIterations),
		(*C.double)(unsafe.Pointer(&x0[0])), 0, (*C.double)(unsafe.Pointer(&out[0])), C.size_t(L), &got); rc != C.GOMEL_OK {
		panic(statusErr(rc))
	}
	return out[:int(got)]
}

// Close releases the device handle (no counterpart in the reference, which holds no resources).
func (m *Mel) Close() {
	m.mu.Lock()
	defer m.mu.Unlock()
	if m.handle != nil {
		C.gomel_destroy(m.handle)
		m.handle = nil
	}
}
