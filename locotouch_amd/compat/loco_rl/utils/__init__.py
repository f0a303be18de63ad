def store_code_state(logdir, repositories):  # reference loco_rl/utils/utils.py: git diffs into the log dir; no GitPython here
    return []


__all__ = ["store_code_state"]
