#!/usr/bin/env python3
"""Lift the reference's golden *output* table into a small committed fixture.

Source (data, not code): ``ChicdiffData/inst/extdata/CD4_Mono_results/test_results.Rds``
and ``test_settings.Rds`` under ``/root/reference`` — the result of one real chr19
``chicdiffPipeline()`` run (24 863 regions, 2v2, norm="combined").
Its *inputs* are missing from the checkout, so it cannot pin counts -> (lfc, p); it
pins the relations the oracle's post-processing must satisfy:

  * pvalue  == 2*pnorm(-|stat|)                (oracle ``oracle_pnorm_two_sided``)
  * stat    == log2FoldChange / lfcSE
  * padj    == BH over the independent-filtering survivors (NA for baseMean below the cutoff)
  * weighted_pvalue == pvalue/weight, weighted_padj == BH(weighted_pvalue)
  * column names / order / dtypes of the ``chicdiffPipeline()`` result

Run here (authoring container only; the reference does not exist on the GPU box):

    python tools/make_golden_from_rds.py

Writes ``tests/golden/chr19_results.npz`` (+ ``chr19_settings.json``).
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
from rds_reader import as_columns, read_rds  # noqa: E402

REF = "/root/reference/ChicdiffData/inst/extdata/CD4_Mono_results"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    res = read_rds(os.path.join(REF, "test_results.Rds"))
    cols = as_columns(res)
    arrays = {k: np.asarray(v) for k, v in cols.items()}
    arrays["__column_order__"] = np.array(list(cols.keys()))
    np.savez_compressed(os.path.join(OUT, "chr19_results.npz"), **arrays)

    st = read_rds(os.path.join(REF, "test_settings.Rds"))
    settings = {}
    for name, v in zip(st.names(), st.value):
        if v is None:
            settings[name] = None
        elif isinstance(v.value, list) and v.value and hasattr(v.value[0], "value"):
            settings[name] = {n: x.names() for n, x in zip(v.names(), v.value)}
        elif isinstance(v.value, np.ndarray):
            settings[name] = [None if (v.value.dtype == np.int32 and x == -2147483648) else x
                              for x in v.value.tolist()]
        else:
            settings[name] = [os.path.basename(s) if isinstance(s, str) and "/" in s else s
                              for s in v.value]
    with open(os.path.join(OUT, "chr19_settings.json"), "w") as fh:
        json.dump(settings, fh, indent=1)
    print("rows", len(arrays["pvalue"]), "cols", len(cols))


if __name__ == "__main__":
    main()

# Design-file fixtures (data only): tests/golden/chr19_HindIII_first3000.rmap is the first 3000 lines of
# ChicdiffData/inst/extdata/designDir/chr19_GRCh37_HindIII.rmap, chr19_baitIDs_first3000.txt the IDs of the
# baits of chr19_GRCh37_HindIII.baitmap that fall in that range (made with head/awk, see git history).
