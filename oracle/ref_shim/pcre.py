"""Stub of the `pcre` package (absent in this image) so /root/reference imports.
Test infrastructure only: used by oracle/make_golden.py and tests that run the
real reference when /root/reference is mounted. Never imported by the product."""
import re as _re
from re import *  # noqa: F401,F403
from re import compile, escape, split, sub, match, search, findall, fullmatch, Pattern  # noqa: F401


class Flag:
    CASELESS = _re.IGNORECASE
    IGNORECASE = _re.IGNORECASE
    MULTILINE = _re.MULTILINE
    DOTALL = _re.DOTALL
    NONE = 0
