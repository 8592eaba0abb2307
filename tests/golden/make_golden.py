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


ML = "/root/reference/errorcorrection/ldpc_examples/matlab_code_Base_matrices/matlab_code & Base_matrices"


def matlab_fixed_point_pins():
    """Statistical pins of the reference's fixed-point layered offset-min-sum decoder (BPSK_nrldpc_sim_RM_FP.m): the FER
    tables in sim_results.m:1-13 (data) and the two NR base matrices they were run on, cut down to the rate-1/2 rate-matched
    part the script decodes (rows 1..mbRM, columns 1..nbRM with nbRM = ceil(kb / Rate) + 2, BPSK_nrldpc_sim_RM_FP.m:8-21),
    written in the AFF3CT .qc layout our readers take."""
    text = open(os.path.join(ML, "sim_results.m")).read()
    pins = {"source": "matlab_code & Base_matrices/sim_results.m:1-13 (BPSK_nrldpc_sim_RM_FP.m, Rate 1/2, rmax 3, MaxItrs 20, offset 2, "
                      "maxqr 31, maxqL 127); columns: EbNo_dB, FER, BER, block errors, bit errors, blocks"}
    for var, name, z, kb in (("res11", "NR_1_1_24", 24, 22), ("res12", "NR_2_6_52", 52, 10)):
        m = re.search(r"%s\s*=\s*\[(.*?)\]" % var, text, re.S)
        rows = [[float(x) for x in line.split()] for line in m.group(1).replace(";", "\n").splitlines() if line.strip()]
        assert len(rows) == 4 and all(len(r) == 6 for r in rows)
        B = [[int(x) for x in line.split()] for line in open(os.path.join(ML, "base_matrices", name + ".txt")) if line.strip()]
        nb_rm = -(-kb * 2 // 1) + 2                      # ceil(kb / 0.5) + 2
        mb_rm = nb_rm - kb
        qc = name + "_rm_half.qc"
        with open(os.path.join(HERE, qc), "w") as f:
            f.write("%d %d %d\n\n" % (nb_rm, mb_rm, z))
            for r in B[:mb_rm]:
                f.write(" ".join(str(x) for x in r[:nb_rm]) + "\n")
        pins[name] = {"qc": qc, "z": z, "kb": kb, "nb_rm": nb_rm, "mb_rm": mb_rm, "rows": rows}
    with open(os.path.join(HERE, "matlab_fp_fer.json"), "w") as f:
        json.dump(pins, f, indent=1)


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
    matlab_fixed_point_pins()
    print("wrote golden fixtures to", HERE)


if __name__ == "__main__":
    sys.exit(main())
