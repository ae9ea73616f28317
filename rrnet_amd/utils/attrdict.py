"""Attribute dictionary standing in for `easydict.EasyDict` (not installed in the image), which
the reference's configs/*.py are written against."""


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        import copy
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})
