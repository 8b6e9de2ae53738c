"""Bisect the fragment-ANI oracle's restatement choices against the reference's fastANI values.

    python tests/tools/fragani_bisect.py [out.md] [--quick]

Scored per variant:
* the 25 rows of tests/golden/{viral,bacterial}_example/fastANI/*.fastani (copies of the reference's fixtures, data
  only): signed dANI (percentage points, ours - fastANI's, after fastANI's six-significant-digit print) and
  d matched per row, the self rows listed separately;
* the pins the reference's tests hold beyond those rows: MIBY01000011 against itself = 99.9953
  (/root/reference/tests/test_self_vs_self.py:121-122), MIBY01000005 against itself == 100
  (/root/reference/tests/test_self_vs_self.py:90-91), and the 3 x 3 identity and coverage matrices of
  /root/reference/tests/test_coverage.py:143-160 (k = 15, fragLen 2000, minFraction 0.15).
CPU only (oracle); rows run on a process pool.  --quick leaves out the bacterial rows (minutes each).
"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from tests.helpers import GOLDEN, read_fasta_bytes  # noqa: E402

K, FRAG = 16, 3000

OLD = {"tie": 0, "float": 0, "freq": 0}  # rounds 1-3 and the first half of round 4: the first of equally good candidates, double sums, every occurrence a seed
VARIANTS = {
    "round 1: seed-implied starts, sketch list 10,60,.., bins (pos+L/2)/L": {"window_rule": 0, "bin_rule": 0, "l2_rule": 0, **OLD},
    "seed-implied starts, window 24, bins pos/(L-20)": {"l2_rule": 0, **OLD},
    "rounds 2-3: slide over reference minimizer starts": {"l2_rule": 1, **OLD},
    "exact slide; position = first minimizer; first candidate on ties; double sums": {"l2_pos": 0, **OLD},
    "exact slide; position = the positions a state stands for; first candidate on ties; double sums (round 4, first half)": {"l2_pos": 1, **OLD},
    "exact slide; position = where the slide arrives at a state; first candidate on ties; double sums": {"l2_pos": 2, **OLD},
    "exact slide, ends with the window's end or past rangeEnd; first candidate on ties; double sums": {"l2_stop": 1, **OLD},
    "exact slide, ends past rangeEnd only; first candidate on ties; double sums": {"l2_stop": 2, **OLD},
    "exact slide; confidence 0.75; first candidate on ties; double sums": {"conf": 0.75, **OLD},
    "exact slide; position = the positions a state stands for; LAST candidate on ties; double sums": {"l2_pos": 1, "tie": 1, "float": 0, "freq": 0},
    "exact slide; position = first minimizer; LAST candidate on ties; double sums": {"tie": 1, "float": 0, "freq": 0},
    "exact slide; position = first minimizer; the candidate libstdc++'s std::sort leaves last; float sums": {"tie": 2, "freq": 0},
    "exact slide; position = first minimizer; last candidate on ties; float sums; every occurrence a seed (no frequency cut)": {"freq": 0},
    "ADOPTED: exact slide; position = first minimizer; LAST candidate on ties; float identities summed in float; Mashmap's frequency cut of the seeds": {},
}
DEFAULTS = {"window_rule": 1, "bin_rule": 1, "l2_rule": 2, "conf": 0.9, "l2_pos": 0, "l2_stop": 0, "tie": 1, "freq": 1, "float": 1}


def contigs_of(path):
    text = read_fasta_bytes(path)
    return [b"".join(rec.split(b"\n")[1:]).translate(None, b" \t\r") for rec in text.split(b">")[1:]]


def printed(ani):
    """fastANI prints the identity with six significant digits"""
    return float(f"{ani:.6g}")


def rows(quick):
    out = []
    for name in ("viral_example",) if quick else ("viral_example", "bacterial_example"):
        for f in sorted((GOLDEN / name / "fastANI").glob("*.fastani")):
            for line in f.read_text().splitlines():
                q, r, ani, matched, total = line.split()
                out.append((name, Path(q).name, Path(r).name, float(ani), int(matched), int(total)))
    return out


def one(args):
    opts, (name, q, r, ani, matched, total) = args
    import oracle

    for key, val in {**DEFAULTS, **opts}.items():
        oracle.fragani_set_option(key, val)
    got_ani, got_m, got_t = oracle.fragani_pair(contigs_of(GOLDEN / name / q), contigs_of(GOLDEN / name / r), K, FRAG, 0.0)
    return (q, r, printed(got_ani) - ani, got_m - matched, total, got_t == total)


def pins(opts):
    import oracle

    for key, val in {**DEFAULTS, **opts}.items():
        oracle.fragani_set_option(key, val)
    small, large = contigs_of(GOLDEN / "MIBY01000005.fasta"), contigs_of(GOLDEN / "MIBY01000011.fasta")
    out = {}
    out["MIBY01000011 self (99.9953)"] = oracle.fragani_pair(large, large, K, FRAG, 0.2)[0]
    out["MIBY01000005 self (100)"] = oracle.fragani_pair(small, small, K, FRAG, 0.2)[0]
    genomes = [small, small + large, large]  # the checksum order of the three files of test_coverage.py
    want_i = [[100.0, 100.0, None], [100.0, 99.997, 99.9959], [None, 99.9959, 99.9959]]
    want_c = [[1.0, 1.0, None], [0.25, 1.0, 0.75], [None, 1.0, 1.0]]
    worst, cov_ok = 0.0, True
    for qi, q in enumerate(genomes):
        for si, s in enumerate(genomes):
            ani, m, t = oracle.fragani_pair(q, s, 15, 2000, 0.15)
            if want_i[qi][si] is None:
                cov_ok &= ani != ani
            else:
                worst = max(worst, abs(printed(ani) - want_i[qi][si]))
                cov_ok &= m / t == want_c[qi][si]
                if (qi, si) in ((1, 1), (1, 2), (2, 2)):
                    out[f"k=15 matrix [{qi}][{si}] ({want_i[qi][si]})"] = ani
    out["k=15 matrix: max |dANI|"] = worst
    out["k=15 matrix: coverage and NULL pattern exact"] = cov_ok
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    quick = "--quick" in sys.argv
    lines = []

    def say(text=""):
        print(text, flush=True)
        lines.append(text)

    all_rows = rows(quick)
    with ProcessPoolExecutor(max_workers=8) as pool:
        for label, opts in VARIANTS.items():
            res = list(pool.map(one, [(opts, row) for row in all_rows]))
            pin = pool.submit(pins, opts).result()
            self_rows = [x for x in res if Path(x[0]).stem.split(".")[0] == Path(x[1]).stem.split(".")[0]]
            other = [x for x in res if x not in self_rows]
            say(f"## {label}")
            say(f"options {opts}")
            say()
            say(f"* all rows: max |dANI| {max(abs(x[2]) for x in res):.4f}, mean {sum(abs(x[2]) for x in res) / len(res):.4f}; "
                f"max |d matched| / total {max(abs(x[3]) / x[4] for x in res) * 100:.2f} %, mean "
                f"{sum(abs(x[3]) / x[4] for x in res) / len(res) * 100:.2f} %; totals exact: {all(x[5] for x in res)}")
            say(f"* rows exactly as fastANI wrote them (identity as printed, kept, total): {sum(1 for x in res if x[2] == 0.0 and x[3] == 0 and x[5])} of {len(res)}")
            say(f"* self rows: {sum(1 for x in self_rows if x[2] == 0.0)} of {len(self_rows)} print as fastANI's; signed dANI "
                + ", ".join(f"{x[2]:+.4f}" for x in self_rows))
            say(f"* other rows: mean signed dANI {sum(x[2] for x in other) / max(1, len(other)):+.4f}")
            say("* pins: " + "; ".join(f"{k} -> {v:.6f}" if isinstance(v, float) else f"{k} -> {v}" for k, v in pin.items()))
            say()
            say("| query | reference | dANI | d matched / total |")
            say("|---|---|---|---|")
            for x in res:
                say(f"| {x[0]} | {x[1]} | {x[2]:+.4f} | {x[3]:+d} / {x[4]} |")
            say()
    if args:
        Path(args[0]).write_text("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
