#!/usr/bin/env python3
"""Checks the transcription SURVEY.md Appendix C -> appendix_c.json: every 16-digit checksum, 8-digit code
word and count in the JSON must occur in the appendix text.  Exit code 0 = consistent."""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def walk(v):
    if isinstance(v, dict):
        for k, x in v.items():
            if not k.startswith("_"):
                yield from walk(x)
    elif isinstance(v, list):
        for x in v:
            yield from walk(x)
    else:
        yield v


def main():
    text = open(os.path.join(ROOT, "SURVEY.md"), encoding="utf-8").read()
    text = text[text.index("## Appendix C"):]
    digits = re.sub(r"(?<=\d)[   ](?=\d{3}\b)", "", text)  # "285 209" -> "285209"
    missing = []
    for v in walk(json.load(open(os.path.join(HERE, "appendix_c.json")))):
        if isinstance(v, str) and re.fullmatch(r"[0-9a-f]{8}|[0-9a-f]{16}", v):
            if v not in text:
                missing.append(v)
        elif isinstance(v, int) and v >= 1000:
            if str(v) not in digits:
                missing.append(v)
    if missing:
        print("not found in SURVEY.md Appendix C:", missing)
        return 1
    print("appendix_c.json is consistent with SURVEY.md Appendix C")
    return 0


if __name__ == "__main__":
    sys.exit(main())
