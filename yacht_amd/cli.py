"""`yacht` command line for the hot-path commands: `yacht train` and `yacht run`
(reference src/yacht/__init__.py:54-136 dispatches the same sub-commands to the same
add_arguments/main pairs), plus `yacht sketch ref|sample` on the HIP sketcher.  The reference's
other sub-commands (download, convert) need the network / NCBI taxonomy tools and are out of this
repository's scope.

    python -m yacht_amd train --ref_file refs.sig.zip --ksize 31 --ani_thresh 0.95 --prefix db --outdir out
    python -m yacht_amd run --json out/db_config.json --sample_file sample.sig.zip --min_coverage_list 1 0.1 --outdir out
"""
from __future__ import annotations

import argparse
import sys

from . import make_training_data_from_sketches, run_YACHT, sketch
from .utils import __version__

OUT_OF_SCOPE = ("download", "convert")


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(prog="yacht", description="YACHT hot path on AMD MI355X (train / run)")
    parser.add_argument("--version", action="version", version=f"yacht-hip (mirrors YACHT {__version__})")
    sub = parser.add_subparsers(dest="command")
    train = sub.add_parser("train", description="Pre-process the reference genomes",
                           formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    make_training_data_from_sketches.add_arguments(train)
    train.set_defaults(func=make_training_data_from_sketches.main)
    run = sub.add_parser("run", description="Run the YACHT algorithm",
                         formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    run_YACHT.add_arguments(run)
    run.set_defaults(func=run_YACHT.main)
    sk = sub.add_parser("sketch", description="Sketch genomes or a sample (DNA FracMinHash, sourmash format)")
    sk_sub = sk.add_subparsers(dest="sketch_command")
    ref = sk_sub.add_parser("ref", description="Sketch reference genomes")
    sketch.add_ref_arguments(ref)
    ref.set_defaults(func=sketch.main_ref)
    smp = sk_sub.add_parser("sample", description="Sketch a metagenomic sample")
    sketch.add_sample_arguments(smp)
    smp.set_defaults(func=sketch.main_sample)
    for name in OUT_OF_SCOPE:
        sub.add_parser(name, add_help=False)
    return parser


def main(argv=None) -> int:
    parser = build_parser()
    argv = sys.argv[1:] if argv is None else argv
    if argv and argv[0] in OUT_OF_SCOPE:
        print(f"`yacht {argv[0]}` is not part of yacht-hip: use the reference YACHT for it", file=sys.stderr)
        return 2
    args = parser.parse_args(argv)
    if not getattr(args, "func", None):
        parser.print_help()
        return 0
    args.func(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
