class ProgressBar:
    def __init__(self, *a, **k):
        pass
