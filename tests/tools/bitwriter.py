"""A tiny MSB-first bit writer with Exp-Golomb codes, for hand-made parameter sets in tests."""


class BitWriter:
    def __init__(self):
        self.bits = []

    def u(self, n, v):
        self.bits += [(v >> i) & 1 for i in range(n - 1, -1, -1)]

    def ue(self, v):
        x = v + 1
        n = x.bit_length() - 1
        self.u(n, 0)
        self.u(n + 1, x)

    def se(self, v):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def trailing(self):
        self.bits.append(1)
        while len(self.bits) % 8:
            self.bits.append(0)

    def bytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))
