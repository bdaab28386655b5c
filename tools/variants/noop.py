pass
