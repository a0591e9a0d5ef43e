#!/usr/bin/env python3
"""Data for the real-data entry (labelany3d_amd/fit_scenes.py), emitted from the reference in THIS container:
``labelany3d_amd/data/coco_category_names.json`` - the id -> name table the reference's annotation reader uses for its
object ids (``COCO_CATEGORIES`` / ``replace_categories_with_supercategories``, reference src/util.py:419-462; unknown ids map to
"unknown").  Data, not code.  No-op where /root/reference is absent.

    python tests/golden/make_golden_scenes.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402


def main():
    ref = _refimport.load()
    if ref is None:
        print("reference not available: nothing written")
        return 0
    table = {str(k): v for k, v in sorted(ref.util.COCO_CATEGORIES.items())}
    # the mapping function is a plain table lookup with "unknown" as the default: pin that on the table itself
    ids = list(ref.util.COCO_CATEGORIES)[:5] + [-1, 12345]
    names = ref.util.replace_categories_with_supercategories(ids)
    assert names[:5] == [table[str(i)] for i in ids[:5]] and names[5:] == ["unknown", "unknown"]
    out = os.path.join(ROOT, "labelany3d_amd", "data", "coco_category_names.json")
    with open(out, "w") as f:
        json.dump(table, f, indent=0)
    print("wrote", out, len(table), "categories")
    return 0


if __name__ == "__main__":
    sys.exit(main())
