"""Stuck time-outs a test provoked on purpose by something other than LENTIL_INJECT_STALL (which the library counts itself):
added up here so that the session's closing assertion (tests/conftest.py) only fails for stalls nobody asked for."""
tolerated = 0
