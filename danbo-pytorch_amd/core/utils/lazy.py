"""A dict whose entries are produced by ONE call on first access: outputs that most callers never read (the loss terms of an
iteration that does not print, the eval-mode `encoded` of DANBO.forward) cost nothing until somebody looks."""


class LazyDict(dict):
    def __init__(self, fill):
        """fill: () -> dict with every entry"""
        super().__init__()
        self._fill_fn = fill

    def _fill(self):
        if self._fill_fn is not None:
            entries = self._fill_fn()          # (a fill that raises stays pending: the next look raises again)
            self._fill_fn = None
            dict.update(self, entries)

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        self._fill()
        return dict.__len__(self)

    def __contains__(self, k):
        self._fill()
        return dict.__contains__(self, k)

    def __bool__(self):
        self._fill()
        return dict.__len__(self) > 0

    def keys(self):
        self._fill()
        return dict.keys(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def get(self, k, default=None):
        self._fill()
        return dict.get(self, k, default)

    def __repr__(self):
        return "LazyDict(<not evaluated>)" if self._fill_fn is not None else dict.__repr__(self)
