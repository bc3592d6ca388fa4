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

from . import phases, train_core, utils
from .utils import logger


# (flag, keyword arguments of add_argument): the reference's argument set (:20-50) + --device
ARGUMENTS = (
    ("--ref_file", dict(required=True,
                        help="Sourmash signature database (.zip with SOURMASH-MANIFEST.csv and signatures/*.sig.gz).")),
    ("--ksize", dict(type=int, required=True, help="k-mer size of the sketches to use.")),
    ("--num_threads", dict(type=int, default=16, help="Host threads for file handling.")),
    ("--ani_thresh", dict(type=float, default=0.95,
                          help='Organisms with this ANI or greater between them are considered "equivalent".')),
    ("--prefix", dict(default="yacht", help="Prefix name to identify this experiment.")),
    ("--outdir", dict(type=str, default=os.getcwd(), help="Path to output directory.")),
    ("--force", dict(action="store_true", help="Overwrite the output directory if it exists.")),
    ("--device", dict(type=int, default=0, help="GPU to run the comparison on.")),
    ("--no_sig_files", dict(action="store_true",
                            help="Do not leave the unzipped signatures/*.sig files in the working directory (the reference does; "
                                 "`yacht run` of this build reads the packed database written next to them instead).")),
    ("--python_ingest", dict(action="store_true",
                             help="Unzip, gunzip and read the signature files in separate passes, as rounds 1-3 did "
                                  "(default: one native pass over the archive).")),
)

# messages the reference raises with (callers and its tests match on them)
MSG_NOT_ZIP = "Reference database file {0} is not a zip file. Please a Sourmash signature database file with Zipfile format."
MSG_NO_ZIP = "Reference database zip file {0} does not exist."
MSG_TMP_EXISTS = ("Temporary directory {0} already exists. Please remove it, use '--force', or given a new prefix name "
                  "using parameter '--prefix'.")
MSG_SCALES = "Not all signatures have the same scaled. Please check your input."


def add_arguments(parser: argparse.ArgumentParser) -> None:
    for flag, kw in ARGUMENTS:
        parser.add_argument(flag, **kw)


def _fresh_workdir(workdir: str, force: bool) -> None:
    if os.path.exists(workdir):
        if not force:
            raise ValueError(MSG_TMP_EXISTS.format(workdir))
        logger.warning(f"Temporary directory {workdir} already exists. Removing it.")
        shutil.rmtree(workdir)
    os.makedirs(workdir, exist_ok=True)


def _extract_members(job) -> int:
    """Extract the listed members; a `.sig.gz` member is inflated on the way and lands as the `.sig` the reference has
    after its unzip + gunzip passes (make_training_data_from_sketches.py:107-133, utils.py:499-509) -- one file created
    per signature instead of two and an unlink, which is what those passes spend their time on at 85 205 members.
    A member that does not inflate is written as it is, for the gunzip pass to complain about."""
    import zlib

    zip_path, workdir, names = job
    made = set()
    root = os.path.normpath(workdir)

    def inside(member: str) -> str:  # the member's path, which must stay inside the working directory (directories too)
        path = os.path.normpath(os.path.join(workdir, member))
        if path != root and not path.startswith(root + os.sep):
            raise ValueError(f"archive member outside the working directory: {member}")
        return path

    def inflate(data: bytes):  # every member of a (possibly concatenated) gzip stream, as gzip / yh_gunzip_files read it
        out = []
        while data:
            d = zlib.decompressobj(31)
            out.append(d.decompress(data))
            out.append(d.flush())
            if not d.eof:
                raise zlib.error("truncated gzip member")
            data = d.unused_data
        return b"".join(out)

    with zipfile.ZipFile(zip_path, "r") as archive:
        for n in names:
            if n.endswith("/"):
                os.makedirs(inside(n), exist_ok=True)
                continue
            data = archive.read(n)
            out = n
            if n.endswith(".sig.gz"):
                try:
                    data = inflate(data)
                    out = n[:-3]
                except zlib.error:
                    pass
            path = inside(out)
            d = os.path.dirname(path)
            if d not in made:
                os.makedirs(d, exist_ok=True)
                made.add(d)
            with open(path, "wb") as f:
                f.write(data)
    return len(names)


def _unpack_database(zip_path: str, workdir: str, threads: int) -> None:
    logger.info("Unzipping the sourmash signature file to the temporary directory")
    with phases.phase("unzip"):
        with zipfile.ZipFile(zip_path, "r") as archive:
            names = archive.namelist()
            if threads > 1 and len(names) > 2000:  # tens of thousands of small members: several readers of the one archive
                from multiprocessing import Pool

                workers = min(threads, 32)
                per = (len(names) + 4 * workers - 1) // (4 * workers)
                jobs = [(zip_path, workdir, names[i:i + per]) for i in range(0, len(names), per)]
                with Pool(workers) as pool:
                    done = sum(pool.imap_unordered(_extract_members, jobs))
                assert done == len(names)
            else:
                _extract_members((zip_path, workdir, names))
    packed = glob.glob(f"{workdir}/signatures/*.sig.gz")  # (only members that did not inflate are left as .gz)
    logger.info(f"Decompressing {len(packed)} .sig.gz files using {threads} threads.")
    with phases.phase("gunzip"):
        utils.decompress_all_sig_files(packed, threads)


def main(args) -> None:
    zip_path = str(Path(args.ref_file).absolute())
    outdir = str(Path(args.outdir).absolute())
    workdir = os.path.join(outdir, args.prefix + "_intermediate_files")

    logger.info("Checking reference database file")
    if os.path.splitext(zip_path)[1] != ".zip":
        raise ValueError(MSG_NOT_ZIP.format(zip_path))
    utils.check_file_existence(zip_path, MSG_NO_ZIP.format(zip_path))
    _fresh_workdir(workdir, args.force)
    extraction = None
    if not zipfile.is_zipfile(zip_path):
        raise zipfile.BadZipFile(f"File is not a zip file: {zip_path}")
    if getattr(args, "python_ingest", False):
        _unpack_database(zip_path, workdir, args.num_threads)
        logger.info("Extracting signature information")
        with phases.phase("signature_metadata"):
            sig_info = utils.collect_signature_info(args.num_threads, args.ksize, workdir)
    else:
        # one pass over the archive: members inflated and parsed by native threads; unless --no_sig_files, the unzipped
        # members are left in the working directory by threads of their own WHILE the comparison runs (85 205 file
        # creations are the directory's lock, not CPU), waited for before the command returns
        logger.info("Reading the sourmash signature database")
        from ._lib import YachtHipError

        try:
            if not getattr(args, "no_sig_files", False):
                extraction = utils.BackgroundExtraction(zip_path, workdir, max(2, min(int(args.num_threads), 16)))
            with phases.phase("ingest"):
                sig_info = utils.ingest_zip_database(zip_path, workdir, args.ksize, args.num_threads, write_files=False,
                                                     background=extraction)
        except YachtHipError as exc:
            # The native reader takes what sourmash writes (stored / deflated members, zip64, CRC-checked); an archive it
            # refuses -- another compression method, a layout it does not know -- gets Python's zipfile's opinion
            # (ADVICE r04): the three-pass route of --python_ingest, which raises BadZipFile itself where the archive IS bad.
            logger.warning(f"the native archive reader refused {zip_path} ({exc}); reading it with Python's zipfile instead")
            train_core.drop_parsed_sketches()
            if extraction is not None:
                try:
                    extraction.wait()
                except Exception:  # noqa: BLE001 -- (it read the same archive: its verdict is the one above)
                    pass
                extraction = None
            _fresh_workdir(workdir, True)
            _unpack_database(zip_path, workdir, args.num_threads)
            logger.info("Extracting signature information")
            with phases.phase("signature_metadata"):
                sig_info = utils.collect_signature_info(args.num_threads, args.ksize, workdir)
    scaled_values = {record[-2] for record in sig_info.values()}
    if len(scaled_values) != 1:
        train_core.drop_parsed_sketches()  # (the ingest's offer must not outlive a run that stops here)
        raise ValueError(MSG_SCALES)

    logger.info("Finding the closely related genomes with ANI > ani_thresh and removing them.")
    try:
        with phases.phase("train_core"):
            kept = utils.run_yacht_train_core(args.num_threads, args.ani_thresh, args.ksize, workdir, sig_info,
                                              device=getattr(args, "device", 0))
    finally:
        train_core.drop_parsed_sketches()

    manifest_path = os.path.join(outdir, f"{args.prefix}_processed_manifest.tsv")
    kept.to_csv(manifest_path, sep="\t", index=None)
    config = {"manifest_file_path": manifest_path, "intermediate_files_dir": workdir, "scale": scaled_values.pop(),
              "ksize": args.ksize, "ani_thresh": args.ani_thresh}
    with open(os.path.join(outdir, f"{args.prefix}_config.json"), "w") as out:
        json.dump(config, out, indent=4)
    if extraction is not None:
        with phases.phase("wait_for_sig_files"):
            extraction.wait()
    logger.info(f"{len(kept)} of {len(sig_info)} references kept; config written to {outdir}")


if __name__ == "__main__":
    p = argparse.ArgumentParser(description="Build a deduplicated reference set from a signature database.",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    add_arguments(p)
    main(p.parse_args())
