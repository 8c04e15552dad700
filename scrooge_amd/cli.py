"""Command-line front door: align the candidate locations of a read set on a GPU.

    python -m scrooge_amd.cli --reference=genome.fa --reads=reads.fastq --seeds=seeds.paf \\
        [--out=aln.paf] [--format=paf|sam] [--reverse_strand] [--read_length_cap=N] \\
        [--dataset_inflation=K] [--W=64 --O=33] [--device=0] [--validate]

Same inputs and preparation as the reference's performance harness
(`tests --reference= --reads= --seeds=`, src/tests.cu:335-410, 782-813: forward-strand candidates,
optional length cap and inflation, reads sorted longest first) and the same report lines, which the
reference's sweep driver scrapes (scripts/profile.py:170-175)."""
import argparse
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="scrooge_amd.cli", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", required=True, help="genome FASTA")
    ap.add_argument("--reads", required=True, help="reads FASTQ")
    ap.add_argument("--seeds", required=True, help="candidate locations, .paf or .maf")
    ap.add_argument("--out", help="write one alignment per candidate (PAF with cg:Z:, or SAM)")
    ap.add_argument("--format", choices=["paf", "sam"], default="paf")
    ap.add_argument("--reverse_strand", action="store_true",
                    help="align '-' candidates with the reverse-complemented read (the reference drops them)")
    ap.add_argument("--read_length_cap", type=int, default=-1)
    ap.add_argument("--dataset_inflation", type=int, default=1)
    ap.add_argument("--W", type=int, default=64)
    ap.add_argument("--O", type=int, default=33)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--validate", action="store_true", help="check every CIGAR against the sequences (validateCigarString)")
    args = ap.parse_args(argv)

    import scrooge_amd
    from scrooge_amd import io as sio

    t0 = time.time()
    job = sio.Job(args.reference, args.reads, args.seeds, reverse_strand=int(args.reverse_strand),
                  read_length_cap=args.read_length_cap, inflation=args.dataset_inflation)
    print("loaded %d reads, %d candidate locations, %d bp reference (%d sequences) in %.1fs"
          % (job.n_reads, job.n_pairs, job.genome_len, job.n_chromosomes, time.time() - t0), file=sys.stderr)
    al = scrooge_amd.Aligner(args.device)
    t1 = time.time()
    alns = job.align(al, out_path=args.out, fmt=args.format, W=args.W, O=args.O)
    wall_ms = (time.time() - t1) * 1e3
    kernel_ms = al.last_timing["kernel_ns"] / 1e6
    # report lines as src/tests.cu:402-406
    print("align_all() took %dms (data transfers, conversion, gpu kernel and post-processing)" % wall_ms)
    print("GPU kernel took %dms" % kernel_ms)
    print("GPU kernel ran at %d aligns/second" % (len(alns) / max(kernel_ms, 1e-6) * 1e3))
    rc = 0
    if args.validate:
        genome, reads, cands, _ = job.views()
        comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
        k = bad = 0
        for r, cs in zip(reads, cands):
            for start, rev in cs:
                q = r.translate(comp)[::-1] if rev else r
                # the alignment consumes a prefix of the suffix: at most len(read) + edits <= 2 * len(read) characters of it
                if sio.validate_alignment(genome[start:start + 2 * len(q) + args.W], q, alns[k].cigar, alns[k].edit_distance) != 0:
                    print("FAILED sanity check for alignment %d" % k)
                    bad += 1
                k += 1
        print("validated %d alignments, %d failed" % (k, bad))
        rc = 1 if bad else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
