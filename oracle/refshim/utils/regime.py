class Regime(object):
    def __init__(self, regime, defaults=None):
        self.regime, self.defaults = regime, defaults or {}
