"""`yacht train` driver: reference .sig.zip -> deduplicated reference set on disk.

Same arguments, same checks and error messages, same artefacts as the reference's
src/yacht/make_training_data_from_sketches.py (:20-155): `{prefix}_intermediate_files/`
(unzipped + gunzipped signatures, training_sig_files.tsv, selected_result.tsv, comparison_files/),
`{prefix}_processed_manifest.tsv` and `{prefix}_config.json`.  The comparison runs on the HIP
engine (utils.run_yacht_train_core); one extra argument, --device, picks the GPU.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import shutil
import zipfile
from pathlib import Path

from . import utils
from .utils import logger


def add_arguments(parser: argparse.ArgumentParser) -> None:
    parser.add_argument("--ref_file", required=True,
                        help="Sourmash signature database (.zip with SOURMASH-MANIFEST.csv and signatures/*.sig.gz).")
    parser.add_argument("--ksize", type=int, required=True, help="k-mer size of the sketches to use.")
    parser.add_argument("--num_threads", type=int, default=16, help="Host threads for file handling.")
    parser.add_argument("--ani_thresh", type=float, default=0.95,
                        help='Organisms with this ANI or greater between them are considered "equivalent".')
    parser.add_argument("--prefix", default="yacht", help="Prefix name to identify this experiment.")
    parser.add_argument("--outdir", type=str, default=os.getcwd(), help="Path to output directory.")
    parser.add_argument("--force", action="store_true", help="Overwrite the output directory if it exists.")
    parser.add_argument("--device", type=int, default=0, help="GPU to run the comparison on.")


def main(args) -> None:
    ref_file = str(Path(args.ref_file).absolute())
    outdir = str(Path(args.outdir).absolute())
    ksize, num_threads, ani_thresh, prefix = args.ksize, args.num_threads, args.ani_thresh, args.prefix

    logger.info("Checking reference database file")
    if os.path.splitext(ref_file)[1] != ".zip":
        raise ValueError(f"Reference database file {ref_file} is not a zip file. Please a Sourmash signature database file with Zipfile format.")
    utils.check_file_existence(ref_file, f"Reference database zip file {ref_file} does not exist.")

    path_to_temp_dir = os.path.join(outdir, prefix + "_intermediate_files")
    if os.path.exists(path_to_temp_dir):
        if not args.force:
            raise ValueError(f"Temporary directory {path_to_temp_dir} already exists. Please remove it, use '--force', or given a new prefix name using parameter '--prefix'.")
        logger.warning(f"Temporary directory {path_to_temp_dir} already exists. Removing it.")
        shutil.rmtree(path_to_temp_dir)
    os.makedirs(path_to_temp_dir, exist_ok=True)

    logger.info("Unzipping the sourmash signature file to the temporary directory")
    with zipfile.ZipFile(ref_file, "r") as z:
        z.extractall(path_to_temp_dir)
    gz = glob.glob(f"{path_to_temp_dir}/signatures/*.sig.gz")
    logger.info(f"Decompressing {len(gz)} .sig.gz files using {num_threads} threads.")
    utils.decompress_all_sig_files(gz, num_threads)

    logger.info("Extracting signature information")
    sig_info_dict = utils.collect_signature_info(num_threads, ksize, path_to_temp_dir)
    scales = {v[-2] for v in sig_info_dict.values()}
    if len(scales) != 1:
        raise ValueError("Not all signatures have the same scaled. Please check your input.")
    scale = scales.pop()

    logger.info("Finding the closely related genomes with ANI > ani_thresh and removing them.")
    manifest_df = utils.run_yacht_train_core(num_threads, ani_thresh, ksize, path_to_temp_dir, sig_info_dict,
                                             device=getattr(args, "device", 0))

    manifest_file_path = os.path.join(outdir, f"{prefix}_processed_manifest.tsv")
    manifest_df.to_csv(manifest_file_path, sep="\t", index=None)
    with open(os.path.join(outdir, f"{prefix}_config.json"), "w") as f:
        json.dump({"manifest_file_path": manifest_file_path, "intermediate_files_dir": path_to_temp_dir,
                   "scale": scale, "ksize": ksize, "ani_thresh": ani_thresh}, f, indent=4)
    logger.info(f"{len(manifest_df)} of {len(sig_info_dict)} references kept; config written to {outdir}")


if __name__ == "__main__":
    p = argparse.ArgumentParser(description="Build a deduplicated reference set from a signature database.",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    add_arguments(p)
    main(p.parse_args())
