"""Pairwise alignment -> observation file, the counterpart of the pairwise branch of the reference's
scripts/prepare-alignments.py:77-111 (which needs BioPython; this does not).

    python -m imcoalhmm_amd.prepare examples/example_data.fa out.ziphmm --names hg18,pantro2
    python -m imcoalhmm_amd.prepare in.fa out.imc --names a,b --cache     # packed 2-bit cache

The text output is byte-compatible with the reference's ("%d " per column, prepare-alignments.py:99-105) and
both outputs are accepted by ``Forwarder(input_filename, NSYM=3)``.  Triplet/quartet alignments (ILS model)
are out of scope.
"""
import argparse
import ctypes
import os

import numpy as np

from . import _capi


def read_fasta(path):
    """{name: sequence} in file order (name = first word of the header)."""
    seqs, name, parts = {}, None, []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                if name is not None:
                    seqs[name] = "".join(parts)
                name, parts = line[1:].split()[0], []
            else:
                parts.append(line)
    if name is not None:
        seqs[name] = "".join(parts)
    return seqs


def read_phylip(path):
    """Sequential or interleaved PHYLIP with names in the first 10 characters / first word."""
    with open(path) as f:
        lines = [l.rstrip("\n") for l in f if l.strip()]
    n, length = (int(x) for x in lines[0].split()[:2])
    names, seqs = [], []
    for l in lines[1:1 + n]:
        parts = l.split(None, 1)
        names.append(parts[0])
        seqs.append(parts[1].replace(" ", "") if len(parts) > 1 else "")
    rest = lines[1 + n:]
    for k, l in enumerate(rest):
        seqs[k % n] += l.replace(" ", "")
    out = dict(zip(names, seqs))
    for k, v in out.items():
        if len(v) != length:
            raise ValueError("PHYLIP sequence %s has %d columns, header says %d" % (k, len(v), length))
    return out


def encode_pairwise(seq1, seq2):
    """uint8 symbols per column: 2 = either base not in ACGT, 0 = equal, 1 = different
    (scripts/prepare-alignments.py:99-105)."""
    if len(seq1) != len(seq2):
        raise ValueError("sequences differ in length: %d vs %d" % (len(seq1), len(seq2)))
    a, b = seq1.encode("ascii"), seq2.encode("ascii")
    out = np.empty(len(a), dtype=np.uint8)
    _capi.check(_capi.lib().imc_encode_pairwise(a, b, len(a), out.ctypes.data_as(_capi._u8p)))
    return out


def write_text(path, obs):
    """The reference's own file format: one decimal token per column, each followed by a space."""
    obs = np.ascontiguousarray(obs, dtype=np.uint8)
    lut = np.array([b"%d " % k for k in range(256)], dtype="S4")
    with open(path, "wb", 64 * 1024) as f:
        for off in range(0, obs.size, 1 << 22):
            f.write(b"".join(lut[obs[off:off + (1 << 22)]].tolist()))


def write_cache(path, obs, nsym=3):
    obs = np.ascontiguousarray(obs, dtype=np.uint8)
    _capi.check(_capi.lib().imc_write_cache(os.fsencode(path), obs.ctypes.data_as(_capi._u8p), obs.size, int(nsym)))


def read_observations(path, nsym=3):
    """Parse a text or cache observation file on the host (what Forwarder.__init__ does before the upload)."""
    n = ctypes.c_size_t(0)
    L = _capi.lib()
    _capi.check(L.imc_read_observations(os.fsencode(path), int(nsym), None, 0, ctypes.byref(n)))
    out = np.empty(n.value, dtype=np.uint8)
    _capi.check(L.imc_read_observations(os.fsencode(path), int(nsym), out.ctypes.data_as(_capi._u8p), out.size, ctypes.byref(n)))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description="pairwise alignment -> IMCoalHMM observation file")
    ap.add_argument("in_filename")
    ap.add_argument("output_filename")
    ap.add_argument("--names", default=None, help="comma-separated pair of sequence names (default: the first two)")
    ap.add_argument("--in-format", default="fasta", choices=["fasta", "phylip"])
    ap.add_argument("--cache", action="store_true", help="write the packed 2-bit cache instead of text")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args(argv)
    seqs = read_fasta(args.in_filename) if args.in_format == "fasta" else read_phylip(args.in_filename)
    names = args.names.split(",") if args.names else list(seqs.keys())[:2]
    if len(names) != 2:
        ap.error("exactly two sequence names are needed for a pairwise alignment")
    obs = encode_pairwise(seqs[names[0]], seqs[names[1]])
    if args.cache:
        write_cache(args.output_filename, obs, 3)
    else:
        write_text(args.output_filename, obs)
    if args.verbose:
        print("%s vs %s: %d columns, symbol counts %s" % (names[0], names[1], obs.size, np.bincount(obs, minlength=3).tolist()))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
