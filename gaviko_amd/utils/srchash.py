"""Hash of what the PRODUCT build compiles from a set of kernel sources.

Measurement records (profiles/r0N_pmc_traffic.json) are only valid for the code they were taken on, so `bench.py` attaches them
when this hash matches.  The product library is compiled without GVK_DIAG (gaviko_amd/build.py): text inside `#ifdef GVK_DIAG`
blocks, comments and blank space never reach it, so none of them may change the hash -- a diag-only edit once withheld
`roofline.traffic` from a driver line although the product's code generation was untouched.
"""
from __future__ import annotations

import hashlib
import os
import re

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
_COMMENT = re.compile(r"//[^\n]*|/\*.*?\*/", re.S)
_DIAG_OPEN = re.compile(r"^\s*#\s*(?:ifdef\s+GVK_DIAG\b|if\s+defined\s*\(?\s*GVK_DIAG\s*\)?\s*$)")
_NDIAG_OPEN = re.compile(r"^\s*#\s*(?:ifndef\s+GVK_DIAG\b|if\s+!\s*defined\s*\(?\s*GVK_DIAG\s*\)?\s*$)")
_IF = re.compile(r"^\s*#\s*if")
_ELSE = re.compile(r"^\s*#\s*else\b")
_ENDIF = re.compile(r"^\s*#\s*endif\b")


def product_text(text: str) -> str:
    """`text` as the product build sees it: comments removed, `#ifdef GVK_DIAG` branches dropped (their `#else` kept, `#ifndef GVK_DIAG`
    the other way round), whitespace runs collapsed, empty lines dropped.  Other conditionals pass through untouched."""
    text = _COMMENT.sub(" ", text)
    out, stack = [], []                    # stack entries: None = a foreign #if; True / False = a GVK_DIAG conditional, branch kept?
    for line in text.split("\n"):
        if _DIAG_OPEN.match(line):
            stack.append(False)
            continue
        if _NDIAG_OPEN.match(line):
            stack.append(True)
            continue
        if _IF.match(line):
            stack.append(None)
        elif _ELSE.match(line) and stack and stack[-1] is not None:
            stack[-1] = not stack[-1]
            continue
        elif _ENDIF.match(line) and stack:
            if stack.pop() is not None:
                continue
        if any(k is False for k in stack):
            continue
        line = " ".join(line.split())
        if line:
            out.append(line)
    return "\n".join(out)


def product_source_hash(files, csrc: str = _CSRC) -> str:
    """First 16 hex digits of the sha256 over the product text of `files` (names relative to gaviko_amd/csrc), in the order given."""
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(csrc, f), "r") as fh:
            h.update(f.encode() + b"\0" + product_text(fh.read()).encode() + b"\0")
    return h.hexdigest()[:16]


GEMM_SOURCES = ("gemm_bf16.hip", "gemm8p_bf16.hip", "gemm_epilogue.hpp", "common.hpp")


def gemm_source_hash() -> str:
    return product_source_hash(GEMM_SOURCES)
