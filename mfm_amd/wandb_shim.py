"""No-op stand-in for ``wandb`` (not installed; logging is outside the hot path): keeps the last run's log in memory."""
history = []
config = None


def init(**kw):
    global config
    history.clear()
    config = kw.get("config")


def log(d):
    history.append(dict(d))


def finish():
    pass


class Table:
    def __init__(self, columns, data):
        self.columns, self.data = columns, data


class Image:
    def __init__(self, fig):
        pass
