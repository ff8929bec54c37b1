import hashlib
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


KEY = 0xCF222F1FE0748978
