"""Emitter for the 1-D-conv networks (``CompactEncoder`` / ``CompactDecoder``)."""


class CompactNet:
    def __init__(self, module, kind, eng):
        raise NotImplementedError("compact (conv) networks: HIP conv kernels not wired yet")
