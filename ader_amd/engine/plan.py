"""placeholder"""


class _Native:
    pass
