#!/usr/bin/env python
"""Builds a PATCHED COPY of the reference's Saber library with the MI355X target added — the saber half of
docs/Manual/addCustomDevice.md:59-280 — in a scratch directory (default integration/_build/anakin, git-ignored).

    python integration/apply_mi355x_target.py [/root/reference] [integration/_build/anakin]

Nothing of the reference is stored in this repository: the script copies `saber/`, `utils/` and the test helper header
from the reference tree where it lies, then makes the one-to-five-line insertions below (each located by a short
anchor string) and drops in this repository's own files:

    saber/saber_types.h                      enum eMI355X + typedef TargetType<eMI355X> MI355X
    saber/core/target_traits.h               struct __mi355x_device + TargetTypeTraits<MI355X> (device target)
    saber/core/target_wrapper.h              #include of impl/mi355x/mi355x_target_wrapper.h (as the MLU target does, :691-693)
    saber/funcs/timer.h                      #include of impl/mi355x/mi355x_timer.h
    saber/funcs/conv.h, conv_eltwise.h       facade ladders: #include of impl/mi355x/saber_conv[_eltwise].h (pattern conv.h:23-50)
    + saber/core/impl/mi355x/{mi355x_target_wrapper.h, mi355x_impl.cpp}, saber/funcs/impl/mi355x/{saber_conv.h,
      mi355x_timer.h}                        (integration/mi355x/*: TargetWrapper / Device / SaberTimer on HIP, SaberConv2D)

The framework half of the manual (:281-455: Net<MI355X>, operator registration, model parser) needs protobuf and a model
and is out of scope here."""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def insert(path, anchor, text, after=True, count=1):
    s = open(path).read()
    assert s.count(anchor) >= 1, (path, anchor)
    i = s.index(anchor) if count == 1 else s.rindex(anchor)
    pos = i + len(anchor) if after else i
    open(path, "w").write(s[:pos] + text + s[pos:])


def main(ref, dst):
    if os.path.exists(dst):
        shutil.rmtree(dst)
    os.makedirs(dst)
    for d in ("saber", "utils"):
        shutil.copytree(os.path.join(ref, d), os.path.join(dst, d),
                        ignore=shutil.ignore_patterns("*.cu", "*.cl", "*.S", "arm", "mlu", "bm", "amd", "cuda"))
    S = os.path.join(dst, "saber")
    insert(os.path.join(S, "saber_types.h"), "    eMLUHX86 = 13,\n", "    eMI355X = 14,\n")
    insert(os.path.join(S, "saber_types.h"), "typedef TargetType<eMLUHX86> MLUHX86;\n",
           "typedef TargetType<eMI355X> MI355X;   // AMD Instinct MI355X (gfx950), HIP runtime\n")
    insert(os.path.join(S, "core", "target_traits.h"), "struct __mlu_device {};\n", "struct __mi355x_device {};\n")
    insert(os.path.join(S, "core", "target_traits.h"), "} //namespace saber",
           "template <>\nstruct TargetTypeTraits<MI355X> {\n    typedef __device_target target_category;\n"
           "    typedef __mi355x_device target_type;\n};\n", after=False)
    insert(os.path.join(S, "core", "target_wrapper.h"), "#endif //ANAKIN_SABER_CORE_TARGET_WRAPPER_H",
           "#ifdef USE_MI355X_PLACE\n#include \"saber/core/impl/mi355x/mi355x_target_wrapper.h\"\n#endif\n\n", after=False)
    insert(os.path.join(S, "funcs", "timer.h"), "#endif //SABER_TIMER_H",
           "#ifdef USE_MI355X_PLACE\n#include \"saber/funcs/impl/mi355x/mi355x_timer.h\"\n#endif\n\n", after=False)
    insert(os.path.join(S, "funcs", "conv.h"), "namespace anakin {",
           "#ifdef USE_MI355X_PLACE\n#include \"saber/funcs/impl/mi355x/saber_conv.h\"\n#endif\n\n", after=False)
    insert(os.path.join(S, "funcs", "conv_eltwise.h"), "namespace anakin {",
           "#ifdef USE_MI355X_PLACE\n#include \"saber/funcs/impl/mi355x/saber_conv_eltwise.h\"\n#endif\n\n", after=False)
    for sub, names in (("core", ("mi355x_target_wrapper.h", "mi355x_impl.cpp")),
                       ("funcs", ("saber_conv.h", "saber_conv_eltwise.h", "mi355x_timer.h"))):
        d = os.path.join(S, sub, "impl", "mi355x")
        os.makedirs(d, exist_ok=True)
        for n in names:
            shutil.copy(os.path.join(HERE, "mi355x", sub, n), d)
    print("patched Saber tree with the MI355X target:", dst)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference",
         sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "_build", "anakin"))
