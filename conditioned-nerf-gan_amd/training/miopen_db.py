"""MIOpen find results shipped with the repo, for the convolution shapes of the GAN step at the benchmarked sizes.

The 3D U-Net encoder and the discriminators are stock PyTorch-ROCm modules (out of the hot path's scope), but their
speed depends on which MIOpen solver runs each convolution: in immediate mode the heuristic picks a naive weight-gradient solver for
several Conv3d shapes (a GAN step at batch 8: 1.65 s), after MIOpen's search (`torch.backends.cudnn.benchmark`) 0.21 s -- and the
search takes 1.5 to 8 minutes per configuration on a fresh machine.  MIOpen keeps what it found in two text files of its user
database directory (`*.ufdb.txt`: ranked solvers per problem, `*.udb.txt`: tuning parameters per solver); the files next to this
module are those of `scripts/make_miopen_db.py` run on an MI355X of this image (gfx950, 256 CUs; MIOpen 3.5.0).  use_shipped_db()
merges them into the process's user database before the first convolution, so that find mode answers from the database instead
of searching.  Problems it does not know are searched as usual (and stay in the user database).  Pure data: `key=value` lines."""
import glob
import os
import shutil

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def use_shipped_db(user_db=None, merge=True):
    """Call before the first convolution of the process.  Returns the user database directory (MIOPEN_USER_DB_PATH).
    merge=False only points the process at the directory: in a multi-rank job ONE rank merges the files (rank 0 first:
    training/latch.py), the others must not write them at the same time."""
    user_db = user_db or os.environ.get("MIOPEN_USER_DB_PATH") or os.path.join(os.path.expanduser("~"), ".config", "miopen")
    os.environ["MIOPEN_USER_DB_PATH"] = user_db
    os.makedirs(user_db, exist_ok=True)
    if not merge:
        return user_db
    for src in glob.glob(os.path.join(HERE, "*.txt")):
        dst = os.path.join(user_db, os.path.basename(src))
        if not os.path.exists(dst):
            shutil.copyfile(src, dst)
            continue
        have = {}
        with open(dst) as f:            # the machine's own results win; the shipped ones fill in what it has not searched yet
            for line in f:
                if "=" in line:
                    have[line.split("=", 1)[0]] = line
        add = []
        with open(src) as f:
            for line in f:
                if "=" in line and line.split("=", 1)[0] not in have:
                    add.append(line if line.endswith("\n") else line + "\n")
        if add:
            with open(dst, "a") as f:
                f.writelines(add)
    return user_db
