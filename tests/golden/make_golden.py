#!/usr/bin/env python3
"""Extract the reference's only exact pins for the LDPC path into data fixtures.

Runs ONLY in the build container (needs /root/reference).  It reads, as text,
the commented-out known-answer vectors in
  BS/src/variants (copy out as main.cpp to use)/main.cpp (alist-v1.0.1):445,447,456,460
(`data`, `encoded`, `llrs`, `decoded`) and copies the H-matrix *data files* the
reference's harness uses.  Nothing here is reference source code: the outputs
are integer / float vectors (JSON) and matrix data files.
"""
import json, os, re, shutil, sys

REF = "/root/reference/errorcorrection/ldpc_examples/my_project_with_aff3ct/examples/bootstrap"
SRC = os.path.join(REF, "src", "variants (copy out as main.cpp to use)", "main.cpp (alist-v1.0.1)")
HERE = os.path.dirname(os.path.abspath(__file__))


def grab(text, name):
    m = re.search(r"std::vector<\w+>\s+%s\s*\{([^}]*)\}" % name, text)
    if not m:
        raise SystemExit("vector %s not found" % name)
    return [float(x) if "." in x else int(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]


def main():
    text = open(SRC).read()
    kat = {
        "source": "main.cpp (alist-v1.0.1):445,447,456,460; H = matrices/H/PEGReg504x1008.alist; "
                  "decoder 'BP flooding SPA', n_ite=10, enable_syndrome=true, syndrome_depth=1",
        "data": grab(text, "data"),
        "encoded": grab(text, "encoded"),
        "llrs": grab(text, "llrs"),
        "decoded": grab(text, "decoded"),
    }
    assert len(kat["data"]) == 504 and len(kat["encoded"]) == 1008
    assert len(kat["llrs"]) == 1008 and len(kat["decoded"]) == 504
    with open(os.path.join(HERE, "kat_peg504x1008.json"), "w") as f:
        json.dump(kat, f)
    for name in ("PEGReg504x1008.alist", "20.alist", "1998.5.3.2665.alist", "test.qc", "test2.qc",
                 "NR_2_3_112.qc", "NR_1_0_2.qc", "NR_1_7_30.qc"):
        shutil.copyfile(os.path.join(REF, "matrices", "H", name), os.path.join(HERE, name))
    print("wrote golden fixtures to", HERE)


if __name__ == "__main__":
    sys.exit(main())
