"""Test-only stand-in for `easydict` (absent from this image, no network).

Only what the reference's model files touch: attribute access == item access,
recursive wrapping of nested dicts, `.keys()`.  Used ONLY by oracle/ref_import.py
to import the unmodified reference under /root/reference on CPU.
"""


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {})
        d.update(kwargs)
        for k, v in d.items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return EasyDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(EasyDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, EasyDict._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e
