"""`python -m gomel_amd.tomel <wav_base_filename | file.wav | file.flac>` -- the counterpart of the reference's cmd/tomel
(cmd/tomel/main.go:10-61): the same fixed parameters (:24-31), `.flac` through ToMelFlac (:33-41), `.wav` through ToMelWav (:42-50),
anything else taken as a base name `<name>.wav` (:51-59); the PNG is written beside the input as `<argument>.png`; the same messages
and exit codes.  The STFT, the mel warp and the log run on the GPU through the C ABI (gomel_to_mel)."""
import sys


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 2:
        print("Usage: python -m gomel_amd.tomel <wav_base_filename>")
        return 1
    filename = argv[1]
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
    try:
        if filename.endswith(".flac"):
            m.ToMelFlac(filename, filename + ".png")
        elif filename.endswith(".wav"):
            m.ToMelWav(filename, filename + ".png")
        else:
            m.ToMelWav(filename + ".wav", filename + ".png")
    except (gomel_amd.ErrFileNotLoaded, gomel_amd.GomelError, OSError) as err:
        print("Error generating mel spectrogram: %s" % (err,))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
