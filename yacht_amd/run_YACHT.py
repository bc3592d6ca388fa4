"""`yacht run` driver: sample .sig.zip + training config -> presence/absence tables.

Same arguments, checks, coverage-list handling and result layout as the reference's
src/yacht/run_YACHT.py (:24-254): `results/result_all.txt` (tab-separated, every user coverage,
unfiltered) and one table per coverage named `min_coverage{c}` (only organisms with
in_sample_est unless --show_all), plus `raw_result` with --keep_raw.  The tables go to
`results/result.xlsx` (yacht_amd.xlsx: no openpyxl needed) and to `results/sheets/<name>.tsv`.
Kept on purpose: the reference fills the column "num_exclusive_kmers_in_sample_sketch" with the
sample's MEAN ABUNDANCE (run_YACHT.py:159) — downstream tools read it that way.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import zipfile
from pathlib import Path

import pandas as pd

from . import hypothesis_recovery_src as hr
from . import phases
from . import utils
from .utils import logger

RAW_RENAMES = {
    "acceptance_threshold_with_coverage": "acceptance_threshold_wo_coverage",
    "actual_confidence_with_coverage": "actual_confidence_wo_coverage",
    "alt_confidence_mut_rate_with_coverage": "alt_confidence_mut_rate_wo_coverage",
}


# (flag, keyword arguments of add_argument): the reference's argument set (run_YACHT.py:24-72)
ARGUMENTS = (
    ("--json", dict(type=str, required=True, help="Config json written by `yacht train`.")),
    ("--sample_file", dict(required=True, help="Metagenomic sample in .sig.zip format")),
    ("--significance", dict(type=float, default=0.99, help="Minimum probability of individual true negative.")),
    ("--num_threads", dict(type=int, default=16, help="Host threads for file handling.")),
    ("--keep_raw", dict(action="store_true", help="Keep raw results in output file.")),
    ("--show_all", dict(action="store_true", help="Show all organisms (no matter if present) in output file.")),
    ("--min_coverage_list", dict(nargs="+", type=float, default=[1, 0.5, 0.1, 0.05, 0.01],
                                 help="Fractions of a genome's unique k-mers assumed covered by the sample, each in [0, 1].")),
    ("--outdir", dict(type=str, default=os.getcwd(), help="Where the 'results' folder is created.")),
)

# messages the reference raises with (callers and its tests match on them)
MSG_NO_CONFIG = "Config file {0} does not exist. Please run make_training_data_from_sketches.py first."
MSG_BAD_COVERAGE = ("One of values in the min_coverage_list you provided {0} is not between 0 and 1. "
                    "Please check your input.")
MSG_NO_MANIFEST = ("The manifest file {0} does not exist. Please check if you are using the correct json file as input.")
MSG_ZIP_WITHOUT_MANIFEST = ("The input file {0} appears to be missing a manifest associated with it. "
                            "Try running: sourmash sig merge {0} -o <new signature with the manifest present>. "
                            "And then run YACHT using the output of that command.")
MSG_NOT_ONE_SKETCH = ("Expected exactly one signature with ksize {1} in {0}, found {2}. "
                      "Likely you will need to do something like: sourmash sig merge {0} -o <new signature with just one sketch in it>.")
MSG_SCALE_MISMATCH = "Sample scale factor does not equal genome scale factor. Please check your input."


def add_arguments(parser: argparse.ArgumentParser) -> None:
    for flag, kw in ARGUMENTS:
        parser.add_argument(flag, **kw)


def coverage_plan(min_coverage_list):
    """De-duplicated, descending coverages with 1.0 forced in front; has_raw tells whether the user
    asked for 1.0 (run_YACHT.py:175-181)."""
    covs = sorted(set(min_coverage_list), reverse=True)
    has_raw = 1.0 in covs
    if not has_raw:
        covs = [1.0] + covs
    return covs, has_raw


def write_tables(tables, results_folder: str) -> None:
    """`results/result.xlsx` -- one sheet per table, the reference's sheet names (run_YACHT.py:231-254) -- written by
    yacht_amd.xlsx (zipfile + XML: no openpyxl needed to WRITE; `pd.read_excel` reads it where openpyxl exists,
    yacht_amd.xlsx.read_xlsx anywhere), and the same tables as `results/sheets/<name>.tsv`."""
    from . import xlsx

    sheets = os.path.join(results_folder, "sheets")
    os.makedirs(sheets, exist_ok=True)
    for name, df in tables:
        df.to_csv(os.path.join(sheets, f"{name}.tsv"), sep="\t", index=False)
    xlsx.write_xlsx(os.path.join(results_folder, "result.xlsx"), tables)


def main(args) -> None:
    json_file_path = str(Path(args.json).absolute())
    sample_file = str(Path(args.sample_file).absolute())
    outdir = str(Path(args.outdir).absolute())
    results_folder = os.path.join(outdir, "results")
    os.makedirs(results_folder, exist_ok=True)

    utils.check_file_existence(json_file_path, MSG_NO_CONFIG.format(json_file_path))
    with open(json_file_path) as f:
        config = json.load(f)
    manifest_file_path, genome_dir = config["manifest_file_path"], config["intermediate_files_dir"]
    scale, ksize, ani_thresh = config["scale"], config["ksize"], config["ani_thresh"]

    if not os.access(outdir, os.W_OK):
        print(f"Cannot write to the location: {outdir}.\n")
        print("Please check that you have the permission to write to this location. Exiting..\n")
        sys.exit(1)
    for x in args.min_coverage_list:
        if not (0 <= x <= 1):
            raise ValueError(MSG_BAD_COVERAGE.format(x))
    utils.check_file_existence(manifest_file_path, MSG_NO_MANIFEST.format(manifest_file_path))

    logger.info("Loading the manifest file generated from the training data.")
    with phases.phase("load_manifest"):
        manifest = pd.read_csv(manifest_file_path, sep="\t", header=0)
    with zipfile.ZipFile(sample_file, "r") as z:
        if "SOURMASH-MANIFEST.csv" not in z.namelist():
            raise FileNotFoundError(MSG_ZIP_WITHOUT_MANIFEST.format(sample_file))
    with phases.phase("load_sample"):
        try:
            sample_sig = utils.load_signature_with_ksize(sample_file, ksize)
        except ValueError:
            raise ValueError(MSG_NOT_ONE_SKETCH.format(sample_file, ksize, len(sample_file)))
        # (the reference parses the file a second time here, run_YACHT.py:150-152 -> utils.py:89-110; same tuple)
        info = (sample_file, sample_sig.name, sample_sig.md5sum(), sample_sig.minhash.mean_abundance,
                len(sample_sig.minhash), sample_sig.minhash.scaled)
    manifest["num_exclusive_kmers_in_sample_sketch"] = info[3]
    manifest["num_total_kmers_in_sample_sketch"] = utils.get_num_kmers(info[3], info[4], info[5], scale=False)
    manifest["sample_scale_factor"] = info[5]
    manifest["min_coverage"] = 1.0
    if scale != info[5]:
        raise ValueError(MSG_SCALE_MISMATCH)

    covs, has_raw = coverage_plan(args.min_coverage_list)

    # databases trained by old versions list *.sig.gz: decompress them once (run_YACHT.py:184-189)
    listed = glob.glob(f"{genome_dir}/training_sig_files.*")
    if listed:
        df = pd.read_csv(listed[0], sep="\t", header=None)
        if len(df) and "sig.gz" in df[0].values[0]:
            pd.DataFrame([x.replace("sig.gz", "sig") for x in df[0]]).to_csv(listed[0], header=False, index=False)
            utils.decompress_all_sig_files(glob.glob(f"{genome_dir}/signatures/*.sig.gz"), args.num_threads)

    logger.info("Computing hypothesis recovery.")
    with phases.phase("hypothesis_recovery"):
        results = hr.hypothesis_recovery(manifest, (sample_file, sample_sig), genome_dir, covs, scale, ksize,
                                         args.significance, ani_thresh, args.num_threads)
    hr.release_reference_dbs()
    results = [r[[c for c in r.columns if c not in ("md5sum", "sample_scale_factor")]]
               .rename(columns={"genome_scale_factor": "scale_factor"}) for r in results]

    logger.info(f"Saving results to {results_folder}.")
    _t_write = phases.phase("write_results")
    _t_write.__enter__()
    user_results = results if has_raw else results[1:]
    user_covs = covs if has_raw else covs[1:]
    pd.concat(user_results, ignore_index=True).to_csv(os.path.join(results_folder, "result_all.txt"), sep="\t", index=False)
    tables = []
    if args.keep_raw:
        tables.append(("raw_result", results[0].rename(columns=RAW_RENAMES)))
    for cov, df in zip(user_covs, user_results):
        tables.append((f"min_coverage{cov}", df if args.show_all else df[df["in_sample_est"] == True]))  # noqa: E712
    write_tables(tables, results_folder)
    _t_write.__exit__(None, None, None)


if __name__ == "__main__":
    p = argparse.ArgumentParser(description="Presence/absence of reference organisms in a metagenomic sample.",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    add_arguments(p)
    main(p.parse_args())
