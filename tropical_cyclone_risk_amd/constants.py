"""Physical constants on the hot path (reference util/constants.py:7)."""
earth_R = 6.3781 * (10**6)        # mean radius of the earth (m)
