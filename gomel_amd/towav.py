"""`python -m gomel_amd.towav <png_filename> [sample_rate]` -- the counterpart of the reference's cmd/towav (cmd/towav/main.go:10-47):
the same fixed parameters (:30-37), default 44100 Hz (:19, an unparsable rate is 0 as strconv.Atoi's dropped error leaves it, :24),
output beside the input as <png_filename>.wav (:43), the same messages and exit codes.  Runs on the GPU through the C ABI."""
import sys


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 2:
        print("Usage: python -m gomel_amd.towav <png_filename>")
        return 1
    filename = argv[1]
    try:
        frequency = int(argv[2]) if len(argv) > 2 else 44100
    except ValueError:
        frequency = 0
    import gomel_amd
    m = gomel_amd.NewMel()
    m.MelFmin = 0
    m.MelFmax = 16000
    m.YReverse = True
    m.Window = 1280
    m.NumMels = 192
    m.Resolut = 4096
    m.GriffinLimIterations = 2
    m.VolumeBoost = 0.0
    m.SampleRate = frequency
    try:
        m.ToWavPng(filename, filename + ".wav")
    except (gomel_amd.ErrFileNotLoaded, gomel_amd.GomelError, OSError) as err:
        print("Error generating wave from spectrogram: %s" % (err,))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
