"""`python -m hipims_mi -c model.xml`: run one model directory on the HIP engine.

Same options as the reference executable (src/main.cpp:464-567): -c/--config-file, -l/--log-file, -s/--quiet-mode,
-n/--disable-screen (accepted, there is no curses screen here), -m/--mpi-mode (one process per GPU is the strip runner's
job: rejected), -x/--code-dir (accepted and ignored: the kernels are compiled into libhipims_mi.so).  Extras:
--output-format xml|.img|.npy|.asc, --batch N (fixed batch size instead of the autotuner).  There is no CPU fallback: without
the HIP library and a GPU this exits with the engine's error.
"""
import argparse
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m hipims_mi", description=__doc__.split("\n\n")[0])
    ap.add_argument("-c", "--config-file", required=True, help="XML-based configuration file defining the model")
    ap.add_argument("-l", "--log-file", default=None, help="File for model execution log")
    ap.add_argument("-s", "--quiet-mode", action="store_true", help="Disable all requirements for user feedback")
    ap.add_argument("-n", "--disable-screen", action="store_true", help="(accepted; no curses screen here)")
    ap.add_argument("-m", "--mpi-mode", action="store_true", help="(not supported: use the strip runner)")
    ap.add_argument("-x", "--code-dir", default=None, help="(accepted and ignored)")
    ap.add_argument("--output-format", default="xml", choices=["xml", ".img", ".npy", ".asc"],
                    help="xml (default): the driver each <dataTarget format=...> names (HFA -> .img, others -> .asc)")
    ap.add_argument("--batch", type=int, default=0, help="fixed batch size (default: autotuned, as the reference)")
    args = ap.parse_args(argv)
    if args.mpi_mode:
        ap.error("--mpi-mode is not supported; multi-GPU runs use hipims_mi.strips (one process per GPU)")

    from .model import Model
    logf = open(args.log_file, "w") if args.log_file else None

    def log(text):
        if logf:
            logf.write(text + "\n")
            logf.flush()
        if not args.quiet_mode:
            print(text, flush=True)

    m = Model(args.config_file, output_format=args.output_format, log=log)
    if args.batch > 0:
        m.scheme.automatic_queue = False
        m.scheme.queue_addition_size = args.batch
    try:
        outs = m.run()
    finally:
        m.close()
        if logf:
            logf.close()
    if not args.quiet_mode:
        rate = m.scheme.cells_calculated / m.seconds if m.seconds > 0 else 0.0
        print(f"Simulation complete: {len(outs)} output times, {m.scheme.iterations} iterations, "
              f"{rate / 1e6:.1f} Mcell-steps/s (counted the reference's way: cols x rows per iteration, skipped included)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
