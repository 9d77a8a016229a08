"""jittor.dataset for the shim: the reference only subclasses Dataset in files the fixtures do not execute"""


class Dataset:
    def __init__(self, *a, **k):
        pass
