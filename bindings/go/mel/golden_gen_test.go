// golden_gen_test.go -- PINNING KIT: reference vectors for the MI355X port, produced BY THE REFERENCE ITSELF.
//
// The port's CPU oracle (oracle/gomel_oracle.py / .c) restates mel/mel.go and mel/impl.go line by line, but three things it
// needs live in un-vendored third-party code and in the Go runtime, and nothing in this repository pins them: gossp's Hann
// table (stft.New, mel/mel.go:50,145), go-dsp's FFT conventions (fft.FFTReal / fft.IFFT, mel/mel.go:95,116) and math/rand's
// global stream (mel/mel.go:82).  A maintainer with a Go toolchain closes that gap in one command:
//
//	cp <port>/bindings/go/mel/golden_gen_test.go  <gomel checkout>/mel/golden_gen_test.go      (UNMODIFIED reference tree)
//	cd <gomel checkout> && GOMEL_GOLDEN_OUT=/tmp/gomel_ref_vectors.json go test ./mel -run TestDumpGoldenVectors -count=1
//	cp /tmp/gomel_ref_vectors.json <port>/tests/golden/ref/
//	cd <port> && python -m pytest tests/test_reference_vectors.py          (oracle: CPU; HIP path: -m gpu on an MI355X)
//
// It is an INTERNAL test of package mel (it calls the unexported pad, isPadded, domel, undomel, hz_to_mel, mel_to_hz).  It
// only reads; it asserts nothing.  Every section of the JSON falsifies one assumption listed in the port's DESIGN.md section 2.
package mel

import (
	"encoding/json"
	"math"
	"math/cmplx"
	"math/rand"
	"os"
	"runtime"
	"testing"

	"github.com/mjibson/go-dsp/fft"
	"github.com/r9y9/gossp/stft"
)

type goldenCfg struct {
	NumMels, Window, Resolut, GriffinLimIterations int
	MelFmin, MelFmax, TuneMul, TuneAdd            float64
}

type goldenFFT struct {
	N       int
	Input   []float64
	FFTReal [][2]float64 // fft.FFTReal(Input)
	IFFT    [][2]float64 // fft.IFFT(fft.FFTReal(Input))
}

type goldenFromMel struct {
	Cfg      goldenCfg
	Mel      [][2]float64 // log-domain input, BEFORE the call
	X0       []float64    // what rand.Float64() hands ISTFT after rand.Seed(Seed): the initial signal of the call
	Seed     int64
	Wave     []float64
	MelAfter [][2]float64 // the caller's slice after the call (spectral_denormalize side effect, mel/mel.go:143)
}

type goldenToMel struct {
	Cfg   goldenCfg
	Wave  []float64
	Pairs [][2]float64
}

type goldenISTFT struct {
	FrameShift, FrameLen, Iterations int
	Window                           []float64
	SpecIn, SpecOut                  [][][2]float64 // [frame][bin](re, im) before / after the call (mel/mel.go:98-108 mutates it)
	X0                               []float64
	Seed                             int64
	Wave                             []float64
}

type goldenTable struct {
	Cfg       goldenCfg
	In        [][2]float64 // one frame: band (or bin) k holds (k+1, 1000+k)
	Undomel   [][2]float64 // undomel(Resolut/2, NumMels, In, fmin, fmax)
	DomelIn   [][2]float64
	Domel     [][2]float64 // domel(Resolut/2, NumMels, DomelIn, fmin, fmax)
	DomelPanic string      // non-empty when domel panics for this configuration (index out of range)
}

type goldenFile struct {
	Meta         map[string]string
	Windows      map[string][]float64 // "hop x frameLen" -> stft.New(hop, frameLen).Window
	FFT          []goldenFFT
	RandSeed1    []float64 // rand.Seed(1); 16 x rand.Float64()
	RandSeed42   []float64
	HzToMel      [][2]float64
	MelToHz      [][2]float64
	Pad          [][3]int // n, window, len(pad(make([]float64, n), window))
	IsPadded     [][4]int // orig, padded, window, 0/1
	Tables       []goldenTable
	FromMel      []goldenFromMel
	ToMel        []goldenToMel
	ISTFT        []goldenISTFT
	NumFrames    [][4]int // len, hop, frameLen, len(stft.New(hop, frameLen).STFT(make([]float64, len)))
}

func c2(v []complex128) [][2]float64 {
	out := make([][2]float64, len(v))
	for i, z := range v {
		out[i] = [2]float64{real(z), imag(z)}
	}
	return out
}

// deterministic, toolchain-independent pseudo-random numbers (64-bit LCG, top 53 bits): the vectors carry their own inputs anyway
type lcg uint64

func (s *lcg) next() float64 {
	*s = *s*6364136223846793005 + 1442695040888963407
	return float64(uint64(*s)>>11) / 9007199254740992.0
}

func testSignal(n int, sr float64, seed uint64) []float64 {
	r := lcg(seed)
	out := make([]float64, n)
	for i := range out {
		t := float64(i) / sr
		out[i] = 0.3*math.Sin(2*math.Pi*(200*t+300*t*t)) + 0.2*math.Sin(2*math.Pi*1234*t)*math.Sin(2*math.Pi*3*t) + 0.02*(r.next()-0.5)
	}
	return out
}

func melOf(c goldenCfg) *Mel {
	m := NewMel()
	m.NumMels, m.Window, m.Resolut, m.GriffinLimIterations = c.NumMels, c.Window, c.Resolut, c.GriffinLimIterations
	m.MelFmin, m.MelFmax, m.TuneMul, m.TuneAdd = c.MelFmin, c.MelFmax, c.TuneMul, c.TuneAdd
	return m
}

func TestDumpGoldenVectors(t *testing.T) {
	path := os.Getenv("GOMEL_GOLDEN_OUT")
	if path == "" {
		t.Skip("set GOMEL_GOLDEN_OUT=<file.json> to dump the reference vectors")
	}
	g := goldenFile{Meta: map[string]string{"go": runtime.Version(), "goos": runtime.GOOS, "goarch": runtime.GOARCH,
		"godebug": os.Getenv("GODEBUG"), "note": "produced by the UNMODIFIED reference (neurlang/gomel) with its own go.mod dependencies"},
		Windows: map[string][]float64{}}

	// 1. gossp's window table and frame count (mel/mel.go:50,77,92,120,145)
	for _, hw := range [][2]int{{1280, 4096}, {256, 1024}, {256, 2048}, {20, 64}, {3, 8}} {
		s := stft.New(hw[0], hw[1])
		key := itoa(hw[0]) + "x" + itoa(hw[1])
		g.Windows[key] = append([]float64(nil), s.Window...)
		for _, n := range []int{hw[1], hw[1] + hw[0] - 1, hw[1] + hw[0], hw[1] + 7*hw[0] + 1} {
			g.NumFrames = append(g.NumFrames, [4]int{n, hw[0], hw[1], len(s.STFT(make([]float64, n)))})
		}
	}

	// 2. go-dsp's transforms: sign, scaling, index order, power-of-two AND other lengths (Bluestein)
	for _, n := range []int{8, 16, 12, 15, 64, 1000} {
		imp := make([]float64, n)
		imp[1] = 1
		ramp := make([]float64, n)
		for i := range ramp {
			ramp[i] = float64(i) - 0.25*float64(n)
		}
		for _, in := range [][]float64{imp, ramp} {
			f := fft.FFTReal(in)
			g.FFT = append(g.FFT, goldenFFT{N: n, Input: in, FFTReal: c2(f), IFFT: c2(fft.IFFT(f))})
		}
	}

	// 3. math/rand's global stream as ISTFT consumes it (mel/mel.go:81-83)
	rand.Seed(1)
	for i := 0; i < 16; i++ {
		g.RandSeed1 = append(g.RandSeed1, rand.Float64())
	}
	rand.Seed(42)
	for i := 0; i < 16; i++ {
		g.RandSeed42 = append(g.RandSeed42, rand.Float64())
	}

	// 4. scalar helpers and padding (mel/impl.go:298-308, 429-479)
	for _, v := range []float64{0, 1, 7.8125, 700, 1000, 7999.999, 8000, 16000, 22050} {
		g.HzToMel = append(g.HzToMel, [2]float64{v, hz_to_mel(v)})
		g.MelToHz = append(g.MelToHz, [2]float64{v, mel_to_hz(v)})
	}
	for _, w := range []int{256, 1280, 275} {
		for _, n := range []int{1, w, 15*w - 1, 15 * w, 15*w + 1, 16 * w, 16*w + 1, 441000, 441599} {
			p := len(pad(make([]float64, n), w))
			g.Pad = append(g.Pad, [3]int{n, w, p})
			for _, q := range []int{p - 1, p, p + 1} {
				b := 0
				if isPadded(n, q, w) {
					b = 1
				}
				g.IsPadded = append(g.IsPadded, [4]int{n, q, w, b})
			}
		}
	}

	// 5. the mel <-> linear index math, edge rows included (mel/impl.go:310-384)
	cfgs := []goldenCfg{
		{192, 1280, 4096, 2, 0, 16000, 1, 0},  // cmd/towav, cmd/tomel
		{80, 256, 1024, 2, 0, 8000, 1, 0},     // TTS geometry (BASELINE configs[3])
		{160, 256, 2048, 2, 0, 8000, 1, 0},    // NewMel defaults
		{8, 20, 64, 3, 0, 8000, 1, 0},         // toy
		{40, 256, 1024, 2, 50, 7600, 0.5, 0.1}, // fmin > 0, Tune* != identity
	}
	for _, c := range cfgs {
		h := c.Resolut / 2
		tb := goldenTable{Cfg: c}
		for k := 0; k < c.NumMels; k++ {
			tb.In = append(tb.In, [2]float64{float64(k + 1), float64(1000 + k)})
		}
		tb.Undomel = undomel(h, c.NumMels, tb.In, c.MelFmin, c.MelFmax)
		for k := 0; k < h; k++ {
			tb.DomelIn = append(tb.DomelIn, [2]float64{float64(k + 1), float64(5000 + k)})
		}
		func() {
			defer func() {
				if r := recover(); r != nil {
					tb.DomelPanic = "panic"
				}
			}()
			tb.Domel = domel(h, c.NumMels, tb.DomelIn, c.MelFmin, c.MelFmax)
		}()
		g.Tables = append(g.Tables, tb)
	}

	// 6. FromMel end to end, with the x0 it drew (mel/mel.go:142-152)
	for ci, c := range cfgs[:4] {
		for _, it := range []int{0, 1, 2, 5} {
			c.GriffinLimIterations = it
			frames := []int{4, 6, 5, 6}[ci]
			r := lcg(1000 + uint64(ci))
			mel := make([][2]float64, frames*c.NumMels)
			lo, hi := math.Log(1e-5), math.Log(50.0)
			for i := range mel {
				mel[i] = [2]float64{lo + (hi-lo)*r.next(), lo + (hi-lo)*r.next()}
			}
			L := c.Resolut + (frames-1)*c.Window
			seed := int64(1 + ci)
			rand.Seed(seed)
			x0 := make([]float64, L)
			for i := range x0 {
				x0[i] = rand.Float64()
			}
			in := append([][2]float64(nil), mel...)
			rand.Seed(seed)
			wave, _ := melOf(c).FromMel(in)
			g.FromMel = append(g.FromMel, goldenFromMel{Cfg: c, Mel: mel, X0: x0, Seed: seed, Wave: wave, MelAfter: in})
		}
	}

	// 7. ToMel end to end (mel/mel.go:46-74)
	for _, c := range cfgs {
		n := 15*c.Window + 3*c.Window + 17
		if c.Resolut > n {
			n = c.Resolut + 2*c.Window + 5
		}
		w := testSignal(n, 22050, 7)
		var pairs [][2]float64
		func() {
			defer func() { _ = recover() }() // a configuration whose domel indexes past the bins panics in the reference
			pairs, _ = melOf(c).ToMel(append([]float64(nil), w...))
		}()
		g.ToMel = append(g.ToMel, goldenToMel{Cfg: c, Wave: w, Pairs: pairs})
	}

	// 8. the exported ISTFT on a complex spectrogram with arbitrary phases and an inconsistent upper half (mel/mel.go:76-139)
	for _, dims := range [][4]int{{3, 8, 4, 2}, {20, 64, 5, 3}, {256, 1024, 4, 2}} {
		hop, n, frames, it := dims[0], dims[1], dims[2], dims[3]
		s := stft.New(hop, n)
		r := lcg(77)
		spec := make([][]complex128, frames)
		in := make([][][2]float64, frames)
		for i := range spec {
			spec[i] = make([]complex128, n)
			for j := range spec[i] {
				spec[i][j] = cmplx.Rect(0.1+1.9*r.next(), 2*math.Pi*(r.next()-0.5))
			}
			in[i] = c2(spec[i])
		}
		L := n + (frames-1)*hop
		rand.Seed(9)
		x0 := make([]float64, L)
		for i := range x0 {
			x0[i] = rand.Float64()
		}
		rand.Seed(9)
		wave := ISTFT(s, spec, it)
		out := make([][][2]float64, frames)
		for i := range spec {
			out[i] = c2(spec[i])
		}
		g.ISTFT = append(g.ISTFT, goldenISTFT{FrameShift: hop, FrameLen: n, Iterations: it, Window: append([]float64(nil), s.Window...),
			SpecIn: in, SpecOut: out, X0: x0, Seed: 9, Wave: wave})
	}

	f, err := os.Create(path)
	if err != nil {
		t.Fatal(err)
	}
	defer f.Close()
	enc := json.NewEncoder(f)
	if err := enc.Encode(&g); err != nil {
		t.Fatal(err)
	}
	t.Logf("wrote %s", path)
}

func itoa(v int) string {
	if v == 0 {
		return "0"
	}
	s := ""
	for v > 0 {
		s = string(rune('0'+v%10)) + s
		v /= 10
	}
	return s
}
