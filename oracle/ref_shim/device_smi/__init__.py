class Device:
    def __init__(self, *a, **k):
        pass
