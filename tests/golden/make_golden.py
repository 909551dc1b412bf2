"""Generates tests/golden/golden.json from the REFERENCE's own prebuilt libjpeg-turbo 3.1.0 (oracle/_ref,
linked from /root/reference/deps/linux/amd64/lib/libjpeg.a) so that the pinned answers travel to the GPU box
where /root/reference does not exist. Run in the build container:  python tests/golden/make_golden.py

Per fixture (copied from /root/reference/data and /root/reference/testdata into tests/golden/inputs):
  * sha256 of the decoded pixels (BGR / gray) as opencv_decoder_read_data produces them,
  * EXIF orientation,
  * the ThumbHash known answer transcribed from /root/reference/thumbhash_test.go:63-81,
  * sha256 of the 256x256 q85 Fit thumbnail = ref decode -> oracle resize restatement -> ref encode
    (the reference's CPU path for BASELINE configs[0]).
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O  # noqa: E402

THUMBHASH = {  # /root/reference/thumbhash_test.go:63-81
    "sunrise.jpg": "1QcSHQRnh493V4dIh4eXh1h4kJUI",
    "sunset.jpg": "3PcNNYSFeXh/d3eld0iHZoZgVwh2",
    "field.jpg": "3OcRJYB4d3h/iIeHeEh3eIhw+j3A",
    "fall.jpg": "HBkSHYSIeHiPiHh8eJd4eTN0EEQG",
    "street.jpg": "VggKDYAW6lZvdYd6d2iZh/p4GE/k",
    "mountain.jpg": "2fcZFIB3iId/h3iJh4aIYJ2V8g==",
    "coast.jpg": "IQgSLYZ6iHePh4h1eFeHh4dwgwg3",
    "firefox-gray.jpg": "FwgOBwAxOWl4l3aQpFiIN5iHBgAAAAAA",
    "large-sunrise.jpg": "VvYRNQRod3x3B4iHeHhYiHeAeQUo",
}


def main():
    assert O.ref() is not None, "oracle/_ref/libref.so missing: run `make -C oracle` where /root/reference exists"
    out = {}
    d = os.path.join(HERE, "inputs")
    for name in sorted(os.listdir(d)):
        data = open(os.path.join(d, name), "rb").read()
        px = O.ref_jpeg_decode(data)
        info = O.jpeg_info(data)
        thumb = O.transform_jpeg_thumbnail(data, 256, 256, 85, use_ref=True)
        out[name] = {
            "width": int(px.shape[1]), "height": int(px.shape[0]), "channels": int(px.shape[2]),
            "orientation": info["orientation"], "dri": info["dri"],
            "pixels_sha256": hashlib.sha256(px.tobytes()).hexdigest(),
            "thumbhash_b64": THUMBHASH.get(name),
            "thumb256_q85_sha256": hashlib.sha256(thumb).hexdigest(),
            "thumb256_q85_len": len(thumb),
        }
    json.dump(out, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(out), "entries")


if __name__ == "__main__":
    main()
