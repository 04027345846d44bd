"""Generates tests/golden/ref_benchmark_csv_shapes.json from the 25 000 timer CSV files the reference SHIPS under benchmarks/ (the
output of its own Timer and decomposition classes on its authors' clusters): per directory kind (pencil, slab_default, slab_z_then_yx,
slab_y_then_zx, ...) the distinct ORDERED lists of section labels of a block, the file-name pattern, and how many files show each.
Row f3 of SURVEY section 8 (timer CSV compatibility) is then checked against files the reference itself wrote:
tests/test_gpu_cpp_drivers.py compares what tools/pencil / tools/slab leave behind with these shapes."""
import collections
import json
import os
import re

REF = "/root/reference/benchmarks"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def shapes():
    out = {}
    for dirpath, _, files in os.walk(REF):
        kind = os.path.basename(dirpath)
        for f in files:
            if not f.endswith(".csv"):
                continue
            lines = open(os.path.join(dirpath, f), errors="replace").read().split("\n")
            if not lines or not lines[0].startswith(","):
                continue
            labels = []
            for ln in lines[1:]:
                if ln == "":
                    if labels:
                        break
                    continue
                labels.append(ln.split(",")[0])
            nranks = len([c for c in lines[0].split(",") if c != ""])
            e = out.setdefault(kind, {"label_lists": collections.Counter(), "name_fields": collections.Counter(), "header_ok": 0, "files": 0})
            e["label_lists"]["\n".join(labels)] += 1
            e["name_fields"][len(re.sub(r"\.csv$", "", f).split("_"))] += 1
            e["header_ok"] += lines[0] == "," + "".join(f"{r}," for r in range(nranks))
            e["files"] += 1
    res = {}
    for kind, e in sorted(out.items()):
        res[kind] = {"files": e["files"], "files_with_the_rank_header_row": e["header_ok"],
                     "fields_in_the_file_name": {str(k): v for k, v in sorted(e["name_fields"].items())},
                     "label_lists": [{"files": n, "labels": k.split("\n")} for k, n in e["label_lists"].most_common()]}
    return res


if __name__ == "__main__":
    res = shapes()
    path = os.path.join(ROOT, "tests", "golden", "ref_benchmark_csv_shapes.json")
    json.dump(res, open(path, "w"), indent=1)
    for kind, e in res.items():
        print(kind, e["files"], "files,", len(e["label_lists"]), "distinct label lists,", e["fields_in_the_file_name"])
