"""Extracts the nav_msgs/Odometry messages of the reference's resource/test_odom.bag (2729 messages on /odometry/lidar_incremental, the
fixture of test/eskf_odom_test.cpp) into tests/golden/test_odom_bag.npz — DATA ONLY: stamps, poses, pose covariance diagonals — and stores
next to them what the independent twin (oracle/twin_eskf.py) makes of them: the filtered state (pos, quat, vel, omega, acc, alpha) after
every eighth message and after the last one, and the final covariance. Run in the authoring container (needs /root/reference); the GPU box only reads the .npz.

    python tests/golden/make_golden_eskf.py
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import twin_eskf  # noqa: E402


def records(b, pos, end):
    while pos < end:
        hl = struct.unpack_from("<I", b, pos)[0]; pos += 4
        hdr = b[pos:pos + hl]; pos += hl
        dl = struct.unpack_from("<I", b, pos)[0]; pos += 4
        data = b[pos:pos + dl]; pos += dl
        f = {}; i = 0
        while i < len(hdr):
            n = struct.unpack_from("<I", hdr, i)[0]; i += 4
            k, v = hdr[i:i + n].split(b"=", 1); i += n
            f[k.decode()] = v
        yield f, data


def parse_odometry(d):
    i = 0
    seq, sec, nsec = struct.unpack_from("<III", d, i); i += 12
    n = struct.unpack_from("<I", d, i)[0]; i += 4 + n          # frame_id
    n = struct.unpack_from("<I", d, i)[0]; i += 4 + n          # child_frame_id
    pose = struct.unpack_from("<7d", d, i); i += 56
    cov = struct.unpack_from("<36d", d, i); i += 288
    return sec, nsec, pose, [cov[k * 7] for k in range(6)]


def main():
    b = open("/root/reference/resource/test_odom.bag", "rb").read()
    assert b[:13] == b"#ROSBAG V2.0\n"
    msgs = []
    for f, data in records(b, 13, len(b)):
        if f["op"][0] == 5:   # chunk (uncompressed in this bag)
            assert f["compression"] == b"none"
            for g, d in records(data, 0, len(data)):
                if g["op"][0] == 2:
                    msgs.append(parse_odometry(d))
    n = len(msgs)
    stamp = np.array([m[0] + 1e-9 * m[1] for m in msgs]); sec = np.array([m[0] for m in msgs], np.uint32); nsec = np.array([m[1] for m in msgs], np.uint32)
    pose = np.array([m[2] for m in msgs]); covd = np.array([m[3] for m in msgs])
    kf = twin_eskf.PoseESEKF()
    accepted = np.zeros(n, bool); filt = np.zeros((n, 19))
    for k in range(n):
        accepted[k] = kf.process_measurement(stamp[k], pose[k, :3], pose[k, 3:])
        filt[k] = np.concatenate([kf.x[:3], kf.orientation(), kf.x[6:18]])
    out = os.path.join(ROOT, "tests", "golden", "test_odom_bag.npz")
    np.savez_compressed(out, sec=sec, nsec=nsec, pose=pose.astype(np.float64), pose_cov_diag=covd, accepted=accepted, filtered_every8=filt[::8], filtered_last=filt[-1], final_P=kf.P)
    print(n, "messages ->", out, os.path.getsize(out), "bytes; accepted", int(accepted.sum()), "span %.1f s" % (stamp[-1] - stamp[0]))
    fusion_fixture(stamp, pose)


FUSION_N = 400                      # the first messages of the bag
FUSION_MAPPING_AT = (50, 200)       # before these messages the back end publishes (at the stamp of message k - 3)
FUSION_BACKEND_P = (0.5, -0.2, 0.1)
FUSION_BACKEND_RPY = (0.01, -0.02, 0.3)


def fusion_fixture(stamp, pose):
    """TransformFusion over the head of the bag, the twin's view: what tests/cpp/ros_wire_demo.cpp `fusion` must reproduce as odomTopic bytes.
    Stamps are the float64 seconds a ROS header carries (sec + 1e-9 nsec); timer ticks at stamp + 0.02 and + 0.045, rounded to whole
    nanoseconds as ros::Time holds them."""
    from scipy.spatial.transform import Rotation
    tw = twin_eskf.TransformFusion()
    bq = Rotation.from_euler("xyz", FUSION_BACKEND_RPY).as_quat()
    rows = []; fut = []
    tick = 0
    for k in range(FUSION_N):
        if k in FUSION_MAPPING_AT:
            tw.mapping_odometry(stamp[k - 3], np.array(FUSION_BACKEND_P), bq)
        tw.lidar_odometry(stamp[k], pose[k, :3], pose[k, 3:])
        for dt in (0.02, 0.045):
            t = stamp[k] + dt
            sec = np.floor(t); nsec = np.round((t - sec) * 1e9)          # ros::Time::fromSec
            now = sec + 1e-9 * nsec
            w = tw.timer(now)
            if w is not None:
                rows.append(np.concatenate([[tick, now], w["position"], w["orientation"], w["velocity"], [w["speed"], float(w["path_appended"]), w["path_length"]]]))
            tick += 1
        wp = tw.predict_timer()
        if wp:
            fut.append(np.concatenate([[k, len(wp)], wp[-1]["position"], Rotation.from_matrix(wp[-1]["R"]).as_quat()]))
    out = os.path.join(ROOT, "tests", "golden", "test_odom_bag_fusion.npz")
    np.savez_compressed(out, n=FUSION_N, mapping_at=np.array(FUSION_MAPPING_AT), backend_p=np.array(FUSION_BACKEND_P), backend_q=bq, ticks=np.array(rows), future=np.array(fut))
    print("fusion fixture:", len(rows), "published ticks of", tick, "->", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
