"""Test-only stand-in for the 12 symbols of timm==0.5.4 the reference model files import.

timm is pinned by the reference (InvPT/README.md:57, TaskPrompter/README.md:75) but is not
installed here and cannot be fetched.  Everything below is plain torch.nn arithmetic restated
from timm 0.5.4's documented behaviour, so it adds no numerical freedom.
Used ONLY by oracle/ref_import.py.
"""
__version__ = "0.5.4-shim"
