#!/usr/bin/env python3
"""Embeds plan.hpp, vm_core.hpp and kernel_body.inc as C++ raw string literals (includes / pragmas stripped) so that
kernels.hip can hand them to hiprtc together with the code generated for one plan."""
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))


def text(name):
    out = []
    for line in open(os.path.join(here, name), encoding="utf-8"):
        if line.startswith("#include") or line.startswith("#pragma once"):
            continue
        out.append(line)
    return "".join(out)


with open(sys.argv[1], "w", encoding="utf-8") as f:
    for var, name in (("kPlanHpp", "plan.hpp"), ("kVmCoreHpp", "vm_core.hpp"), ("kKernelBody", "kernel_body.inc")):
        f.write("static const char %s[] = R\"GKSRC(%s)GKSRC\";\n" % (var, text(name)))
