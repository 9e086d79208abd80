"""Summary block of a benchmark log in the form the reference's GUI mains print it (thirdparty/perfstats.h:330-343, the closing
loop of ``PerfStats::print_all_data``, called by mainQt.cpp:176 / SlamBenchQt.cpp:282): one line per column,

    # <name>\tMean:\t<mean>\tStandard deviation:\t<sd>\tmin:\t<min>\tmax:\t<max>

Held to the reference's own compiled code: ``tests/golden/perfstats_vectors.json`` is what that header, compiled where it lies with
the reference's ``-O3`` (the probe of ``tests/golden/make_perfstats_vectors.py``), prints for seeded sample tables; this module reproduces it character for
character.  Three things the header does that a reading of it would not suggest:

* the block always goes to ``std::cout``.  When the log itself went to stdout too, that stream has been switched to fixed notation
  and left alignment by the per-frame rows before: ``# name<padded to 13>\t...`` with ten decimals (``log_on_stdout=True``); when the
  log went to a file, stdout is still in its default state: name right-aligned in 13 columns, numbers as ``%.10g``;
* the "standard deviation" is ``sqrt(sum((x - mean)^2) / (N + 1))``: the divisor is the row counter of the loop that printed the rows,
  which runs one pass longer than there are rows (that pass prints the empty line);
* that loop's end condition reads an uninitialised counter (``int count;``, perfstats.h:268-271): with the reference's ``-O3`` it behaves
  as if it started at zero, which is the build this module follows (unoptimised, the rows are skipped and the deviation is NaN).

Input is the TSV log of ``kfusion-benchmark-*`` (benchmark.cpp:120-170); SURVEY.md section 8(f) row 4.

    python -m slambench_amd.perfstats benchmark.log [--log-on-stdout]
"""
from __future__ import annotations

import math
import sys

from . import checkpos


def summary(log_text: str) -> dict:
    cols = checkpos.parse_log(log_text)["columns"]
    return summarise_columns(cols)


def summarise_columns(cols: dict) -> dict:
    out = {}
    for name, v in cols.items():
        if not v:
            continue
        mean = 0.0
        for x in v:                     # perfstats.h:250-258: running sum in row order, divided by the row count
            mean = mean + x
        mean = mean / len(v)
        sd = 0.0
        for x in v:                     # perfstats.h:322-324
            sd = sd + ((x - mean) * (x - mean))
        out[name] = {"mean": mean, "sd": math.sqrt(sd / (len(v) + 1)), "min": min(v), "max": max(v), "n": len(v)}
    return out


def render(stats: dict, log_on_stdout: bool = False) -> str:
    lines = []
    for name, s in stats.items():
        if log_on_stdout:
            lines.append("# %-13s\tMean:\t%.10f\tStandard deviation:\t%.10f\tmin:\t%.10f\tmax:\t%.10f" % (name, s["mean"], s["sd"], s["min"], s["max"]))
        else:
            lines.append("# %13s\tMean:\t%.10g\tStandard deviation:\t%.10g\tmin:\t%.10g\tmax:\t%.10g" % (name, s["mean"], s["sd"], s["min"], s["max"]))
    return "\n".join(lines) + "\n"


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    on_stdout = "--log-on-stdout" in argv
    argv = [a for a in argv if a != "--log-on-stdout"]
    if len(argv) != 1:
        print("usage: python -m slambench_amd.perfstats <benchmark.log> [--log-on-stdout]", file=sys.stderr)
        return 1
    sys.stdout.write(render(summary(open(argv[0]).read()), on_stdout))
    return 0


if __name__ == "__main__":
    sys.exit(main())
