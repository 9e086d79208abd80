"""Summary block of a benchmark log in the form the reference's GUI mains print with ``-q`` (thirdparty/perfstats.h:330-343,
``PerfStats::print_all_data``): one line per column,

    # <name>\tMean:\t<mean>\tStandard deviation:\t<sd>\tmin:\t<min>\tmax:\t<max>

with the population standard deviation sqrt(sum((x - mean)^2) / N) the reference computes.  Input is the TSV log of
``kfusion-benchmark-*`` (benchmark.cpp:120-170); SURVEY.md section 8(f) row 4.

    python -m slambench_amd.perfstats benchmark.log
"""
from __future__ import annotations

import math
import sys

from . import checkpos


def summary(log_text: str) -> dict:
    cols = checkpos.parse_log(log_text)["columns"]
    out = {}
    for name, v in cols.items():
        if name == "frame" or not v:
            continue
        mean = sum(v) / len(v)
        sd = math.sqrt(sum((x - mean) * (x - mean) for x in v) / len(v))
        out[name] = {"mean": mean, "sd": sd, "min": min(v), "max": max(v), "n": len(v)}
    return out


def render(stats: dict) -> str:
    lines = []
    for name, s in stats.items():
        lines.append("# %13s\tMean:\t%.10f\tStandard deviation:\t%.10f\tmin:\t%.10f\tmax:\t%.10f" % (name, s["mean"], s["sd"], s["min"], s["max"]))
    return "\n".join(lines) + "\n"


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        print("usage: python -m slambench_amd.perfstats <benchmark.log>", file=sys.stderr)
        return 1
    sys.stdout.write(render(summary(open(argv[0]).read())))
    return 0


if __name__ == "__main__":
    sys.exit(main())
