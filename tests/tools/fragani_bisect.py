"""Bisect the fragment-ANI oracle's restatement choices against the reference's 25 fastANI rows.

    python tests/tools/fragani_bisect.py [out.md]
For every variant: max and mean |dANI| (percentage points) and max |d matched| / total over the 25 rows of
tests/golden/{viral,bacterial}_example/fastANI/*.fastani (copies of the reference's fixtures, data only).
CPU only (oracle); rows run on a process pool.
"""
import itertools
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from tests.helpers import GOLDEN, read_fasta_bytes  # noqa: E402

K, FRAG = 16, 3000


def contigs_of(path):
    text = read_fasta_bytes(path)
    return [b"".join(rec.split(b"\n")[1:]).translate(None, b" \t\r") for rec in text.split(b">")[1:]]


def rows():
    out = []
    for name in ("viral_example", "bacterial_example"):
        for f in sorted((GOLDEN / name / "fastANI").glob("*.fastani")):
            for line in f.read_text().splitlines():
                q, r, ani, matched, total = line.split()
                out.append((name, Path(q).name, Path(r).name, float(ani), int(matched), int(total)))
    return out


def one(args):
    opts, (name, q, r, ani, matched, total) = args
    import oracle

    for key, val in opts.items():
        oracle.fragani_set_option(key, val)
    got_ani, got_m, got_t = oracle.fragani_pair(contigs_of(GOLDEN / name / q), contigs_of(GOLDEN / name / r), K, FRAG, 0.0)
    return (q, r, got_ani - ani, got_m - matched, total, got_t == total)


def evaluate(pool, opts):
    res = list(pool.map(one, [(opts, row) for row in rows()]))
    d_ani = [abs(x[2]) for x in res]
    d_m = [abs(x[3]) / x[4] for x in res]
    worst = max(res, key=lambda x: abs(x[2]))
    return max(d_ani), sum(d_ani) / len(d_ani), max(d_m), sum(d_m) / len(d_m), all(x[5] for x in res), worst


def main():
    grid = {"window_rule": (0, 1), "bin_rule": (0, 1), "l2_rule": (0, 1), "conf": (0.9, 0.75)}
    lines = ["| window rule | bin rule | L2 rule | conf | max dANI (pp) | mean dANI | max d matched / total | mean | totals exact | worst row |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    with ProcessPoolExecutor(max_workers=8) as pool:
        for combo in itertools.product(*grid.values()):
            opts = dict(zip(grid, combo))
            if opts["conf"] == 0.75 and (opts["l2_rule"] == 1 or opts["bin_rule"] != opts["window_rule"]):
                continue  # the confidence level is bisected on the two corner variants only
            mx, mean, mm, mmean, exact, worst = evaluate(pool, opts)
            line = (f"| {'1,2,5,10,20,..' if opts['window_rule'] else '10,60,110,..'} | {'pos/(L-20)' if opts['bin_rule'] else '(pos+L/2)/L'} | "
                    f"{'slide over ref minimizers' if opts['l2_rule'] else 'seed-implied starts'} | {opts['conf']} | {mx:.4f} | {mean:.4f} | "
                    f"{mm * 100:.2f} % | {mmean * 100:.2f} % | {exact} | {worst[0]} vs {worst[1]} ({worst[2]:+.4f}) |")
            print(line, flush=True)
            lines.append(line)
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
