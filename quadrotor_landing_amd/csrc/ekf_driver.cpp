// ekf_driver.cpp -- ROS-free replacement of the reference's rel_pose_EKF_node executable
// (quad_state_estimation/src/relative_pose_EKF_main.cpp + relative_pose_EKF_node.cpp).
//
// Where the node loads ROS parameters (NODE.cpp:11-136), subscribes to IMU / AprilTag topics
// (NODE.cpp:39-40) and runs filter_update on a timer (NODE.cpp:50,178-182), this driver
//   1. reads the same keys from one of the reference's EKF YAML files,
//   2. creates a batch of filters on one MI355X through the C-ABI,
//   3. generates a seeded synthetic IMU + tag-pose sequence on the device (qle_synth_generate),
//   4. runs the tick loop with the decision logic on the device (rate limit + corner gate),
//   5. prints what the node publishes for filter 0 (NODE.cpp:192-220) and the RMSE vs the truth.
//
//   ekf_driver --config relative_pose_EKF_rotors.yaml --batch 65536 --ticks 1000 [--dtype f32|f64]
//              [--device 0] [--seed N] [--update-freq HZ] [--measurement-freq HZ] [--corner-gate 0|1] [--multirate 0|1]
//              [--json]        print ONE JSON line with the fields of bench.py's line (value, ms_per_step, per-shard device times)
//              [--devices N]   shard the batch over N devices in this process: one handle + one HIP stream per
//                              device, one host thread each, no collective; the three RMSE sums and the reports
//                              are combined on the host.  Shards wrap onto the devices
//                              that exist, so N > #GPUs is a functional rehearsal.
//              [--sequence LOG.csv [--trace OUT.csv]]
//                              replay a RECORDED event log instead of generating one: the node's three callbacks
//                              (NODE.cpp:144-182) are driven from the file in time order --
//                                  imu,<t>,<ax>,<ay>,<az>,<wx>,<wy>,<wz>                       IMUSubCallback
//                                  tag,<t_arrival>,<header stamp>,<px>,<py>,<pz>,<qx>,<qy>,<qz>,<qw>   AprilTagSubCallback
//                              and FilterUpdateCallback fires every 1/update_freq s from the first event on.  Every
//                              filter of the batch sees the same stream; --trace writes what the node would publish
//                              for filter 0 on every tick (NODE.cpp:192-281).
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/qle_ekf.h"

namespace {

// Minimal parser for the subset of YAML the reference's parameter files use:
// `key: scalar`, `key: [a, b, ...]` (lists may span lines), `#` comments.
std::map<std::string, std::vector<std::string>> parse_yaml(const std::string& path)
{
    std::ifstream in(path);
    if (!in) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(2); }
    std::map<std::string, std::vector<std::string>> out;
    std::string line, key, acc;
    bool in_list = false;
    auto strip = [](std::string s) {
        size_t a = s.find_first_not_of(" \t\r\n\"'"), b = s.find_last_not_of(" \t\r\n\"'");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    };
    auto finish_list = [&]() {
        std::vector<std::string> v;
        std::stringstream ss(acc);
        std::string tok;
        while (std::getline(ss, tok, ',')) { tok = strip(tok); if (!tok.empty()) v.push_back(tok); }
        out[key] = v;
        in_list = false; acc.clear();
    };
    while (std::getline(in, line)) {
        size_t h = line.find('#');
        if (h != std::string::npos) line = line.substr(0, h);
        if (strip(line).empty()) continue;
        if (in_list) {
            size_t e = line.find(']');
            acc += " " + (e == std::string::npos ? line : line.substr(0, e));
            if (e != std::string::npos) finish_list();
            continue;
        }
        size_t c = line.find(':');
        if (c == std::string::npos) continue;
        key = strip(line.substr(0, c));
        std::string val = strip(line.substr(c + 1));
        if (!val.empty() && val[0] == '[') {
            size_t e = val.find(']');
            acc = e == std::string::npos ? val.substr(1) : val.substr(1, e - 1);
            in_list = true;
            if (e != std::string::npos) finish_list();
        } else {
            out[key] = {val};
        }
    }
    return out;
}

bool as_bool(const std::string& s) { return s == "True" || s == "true" || s == "1" || s == "yes"; }

void check(int rc, const char* what)
{
    if (rc != QLE_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, qle_last_error()); std::exit(1); }
}

// ---- recorded event log: the ROS side of the node, stubbed ----------------------------------------------------
struct Event {
    double t = 0;       // arrival time: when the node's callback would run
    bool tag = false;   // false: IMU sample (v[0..5]); true: tag detection (stamp + v[0..6] = position, orientation xyzw)
    double stamp = 0;   // AprilTagDetectionArray header stamp (NODE.cpp:167)
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
};

std::vector<Event> read_event_log(const std::string& path)
{
    std::ifstream in(path);
    if (!in) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(2); }
    std::vector<Event> ev;
    std::string line;
    int64_t ln = 0;
    while (std::getline(in, line)) {
        ++ln;
        size_t h = line.find('#');
        if (h != std::string::npos) line = line.substr(0, h);
        if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;
        std::vector<std::string> f;
        std::stringstream ss(line);
        std::string tok;
        while (std::getline(ss, tok, ',')) {
            size_t a = tok.find_first_not_of(" \t\r\n"), b = tok.find_last_not_of(" \t\r\n");
            f.push_back(a == std::string::npos ? std::string() : tok.substr(a, b - a + 1));
        }
        Event e;
        const size_t want = f[0] == "imu" ? 8 : f[0] == "tag" ? 10 : 0;
        if (want == 0 || f.size() != want) { std::fprintf(stderr, "%s:%lld: expected `imu` + 7 numbers or `tag` + 9 numbers\n", path.c_str(), (long long)ln); std::exit(2); }
        e.tag = f[0] == "tag";
        e.t = std::strtod(f[1].c_str(), nullptr);
        size_t k = 2;
        if (e.tag) e.stamp = std::strtod(f[k++].c_str(), nullptr);
        for (int j = 0; k < f.size(); ++k, ++j) e.v[j] = std::strtod(f[k].c_str(), nullptr);
        if (!ev.empty() && e.t < ev.back().t) { std::fprintf(stderr, "%s:%lld: events must be ordered by arrival time\n", path.c_str(), (long long)ln); std::exit(2); }
        ev.push_back(e);
    }
    return ev;
}

// The node's main loop on a recorded log (MAIN.cpp:17 single-threaded spinner): callbacks in arrival order, the
// filter_update timer every dT_nom.  Returns the number of ticks the filter was active for.
int run_recorded(const qle_params& p, const qle_derived& d, const std::string& log, const std::string& trace_path, int64_t B, int dtype, int device)
{
    const std::vector<Event> ev = read_event_log(log);
    if (ev.empty()) { std::fprintf(stderr, "%s holds no events\n", log.c_str()); return 2; }
    qle_batch* h = nullptr;
    check(qle_create(&h, B, dtype, device, &p), "qle_create");
    check(qle_enable_gating(h, 1), "qle_enable_gating");
    check(qle_enable_aux(h, 1), "qle_enable_aux");
    std::vector<double> u((size_t)B * 6, 0.0), z((size_t)B * 7, 0.0), stamp((size_t)B, 0.0);
    std::vector<uint8_t> ready((size_t)B, 0), perf((size_t)B, 0), cons((size_t)B, 0);
    std::vector<int32_t> upds((size_t)B, 0);
    std::vector<double> pose((size_t)B * 7), cov((size_t)B * 36), vel((size_t)B * 3), bias((size_t)B * 6), accel((size_t)B * 3), obs((size_t)B * 7), delay((size_t)B, p.measurement_delay);
    for (int64_t i = 0; i < B; ++i) z[(size_t)i * 7 + 6] = 1.0;
    FILE* tr = nullptr;
    if (!trace_path.empty()) {
        tr = std::fopen(trace_path.c_str(), "w");
        if (!tr) { std::fprintf(stderr, "cannot write %s\n", trace_path.c_str()); return 2; }
        std::fprintf(tr, "# t,rx,ry,rz,qx,qy,qz,qw,vx,vy,vz,ax,ay,az,performed_correction,upds_since_correction,measurement_delay_curr,cov_rr_xx,cov_tt_xx\n");
    }
    bool state_initialized = false, measurement_ready = false;
    int64_t n_ticks = 0, n_active = 0, n_corr = 0;
    size_t k = 0;
    const double t_first = ev.front().t, t_last = ev.back().t;
    for (int64_t tick = 1;; ++tick) {
        const double t = t_first + (double)tick * d.dT_nom;
        if (t > t_last + 0.5 * d.dT_nom) break;
        for (; k < ev.size() && ev[k].t <= t; ++k) {
            const Event& e = ev[k];
            if (!e.tag) {   // IMUSubCallback, NODE.cpp:144-151
                for (int64_t i = 0; i < B; ++i) std::memcpy(&u[(size_t)i * 6], e.v, sizeof(double) * 6);
            } else {        // AprilTagSubCallback, NODE.cpp:153-176
                for (int64_t i = 0; i < B; ++i) { std::memcpy(&z[(size_t)i * 7], e.v, sizeof(double) * 7); stamp[(size_t)i] = e.stamp; }
                measurement_ready = true;
                if (!state_initialized) { check(qle_initialize_state(h, z.data(), 0), "qle_initialize_state"); state_initialized = true; }
            }
        }
        ++n_ticks;
        if (!state_initialized) continue;   // EKF.cpp:129-130
        std::fill(ready.begin(), ready.end(), (uint8_t)(measurement_ready ? 1 : 0));
        check(qle_filter_update_stamped(h, u.data(), measurement_ready ? z.data() : nullptr, measurement_ready ? ready.data() : nullptr, t, stamp.data()), "qle_filter_update_stamped");
        check(qle_get_tick_flags(h, perf.data(), cons.data(), upds.data()), "qle_get_tick_flags");
        if (cons[0]) measurement_ready = false;   // EKF.cpp:152 (every filter sees the same stream, so the flags agree)
        ++n_active;
        n_corr += perf[0];
        if (tr) {
            check(qle_get_report(h, pose.data(), cov.data(), vel.data(), bias.data()), "qle_get_report");
            check(qle_get_aux(h, accel.data(), obs.data()), "qle_get_aux");
            if (perf[0] && p.multirate_ekf) check(qle_get_measurement_delay(h, delay.data()), "qle_get_measurement_delay");
            std::fprintf(tr, "%.9f", t);
            for (int j = 0; j < 7; ++j) std::fprintf(tr, ",%.17g", pose[(size_t)j]);
            for (int j = 0; j < 3; ++j) std::fprintf(tr, ",%.17g", vel[(size_t)j]);
            for (int j = 0; j < 3; ++j) std::fprintf(tr, ",%.17g", accel[(size_t)j]);
            std::fprintf(tr, ",%d,%d,%.17g,%.17g,%.17g\n", (int)perf[0], (int)upds[0], delay[0], cov[0], cov[21]);
        }
    }
    if (tr) std::fclose(tr);
    int64_t bad = 0;
    check(qle_count_nonfinite(h, &bad), "qle_count_nonfinite");
    check(qle_get_report(h, pose.data(), cov.data(), vel.data(), bias.data()), "qle_get_report");
    std::printf("%s\n", qle_version());
    std::printf("recorded log %s: %zu events over %.3f s, %lld timer ticks at %.1f Hz, filter active for %lld, corrections performed %lld, multirate %d\n", log.c_str(),
                ev.size(), t_last - t_first, (long long)n_ticks, p.update_freq, (long long)n_active, (long long)n_corr, p.multirate_ekf);
    std::printf("filter 0: rel_pose position (%.4f, %.4f, %.4f) orientation xyzw (%.4f, %.4f, %.4f, %.4f)\n", pose[0], pose[1], pose[2], pose[3], pose[4], pose[5], pose[6]);
    std::printf("filter 0: velocity (%.4f, %.4f, %.4f)  IMU bias+static accel (%.4f, %.4f, %.4f) gyro (%.5f, %.5f, %.5f); non-finite filters: %lld\n", vel[0], vel[1], vel[2],
                bias[0], bias[1], bias[2], bias[3], bias[4], bias[5], (long long)bad);
    qle_destroy(h);
    return bad == 0 ? 0 : 3;
}

}  // namespace

int main(int argc, char** argv)
{
    std::string config, sequence, trace;
    int64_t batch = 4096, ticks = 1000;
    bool batch_given = false;
    int dtype = QLE_F32, device = 0;
    uint64_t seed = 0xE4F00001ULL;
    double update_freq = 0, measurement_freq = 0;
    int corner_gate = -1, multirate = -1, n_devices = 1;
    bool json = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return std::string(argv[++i]); };
        if (a == "--config") config = next();
        else if (a == "--batch") { batch = std::atoll(next().c_str()); batch_given = true; }
        else if (a == "--sequence") sequence = next();
        else if (a == "--trace") trace = next();
        else if (a == "--ticks") ticks = std::atoll(next().c_str());
        else if (a == "--dtype") dtype = next() == "f64" ? QLE_F64 : QLE_F32;
        else if (a == "--device") device = std::atoi(next().c_str());
        else if (a == "--seed") seed = std::strtoull(next().c_str(), nullptr, 0);
        else if (a == "--update-freq") update_freq = std::atof(next().c_str());
        else if (a == "--measurement-freq") measurement_freq = std::atof(next().c_str());
        else if (a == "--corner-gate") corner_gate = std::atoi(next().c_str());
        else if (a == "--multirate") multirate = std::atoi(next().c_str());
        else if (a == "--devices") n_devices = std::max(1, std::atoi(next().c_str()));
        else if (a == "--json") json = true;
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }

    qle_params p;
    check(qle_params_default(&p), "qle_params_default");
    if (!config.empty()) {
        auto y = parse_yaml(config);
        auto num = [&](const char* k, double& dst) { if (y.count(k)) dst = std::atof(y[k][0].c_str()); };
        auto flag = [&](const char* k, int32_t& dst) { if (y.count(k)) dst = as_bool(y[k][0]) ? 1 : 0; };
        auto vec = [&](const char* k, double* dst, size_t cap, bool required) {
            if (!y.count(k)) { if (required) { std::fprintf(stderr, "%s: required key %s missing (NODE.cpp reads it with getParam, no default)\n", config.c_str(), k); std::exit(2); } return (size_t)0; }
            if (y[k].size() > cap) { std::fprintf(stderr, "%s has %zu values, capacity %zu\n", k, y[k].size(), cap); std::exit(2); }
            for (size_t j = 0; j < y[k].size(); ++j) dst[j] = std::atof(y[k][j].c_str());
            return y[k].size();
        };
        num("update_freq", p.update_freq); num("measurement_freq", p.measurement_freq);          // NODE.cpp:31-32
        num("measurement_delay", p.measurement_delay); num("measurement_delay_max", p.measurement_delay_max);
        num("dyn_measurement_delay_offset", p.dyn_measurement_delay_offset);                      // NODE.cpp:33-35
        flag("limit_measurement_freq", p.limit_measurement_freq);                                 // NODE.cpp:36
        flag("est_bias", p.est_bias); flag("corner_margin_enbl", p.corner_margin_enbl);          // NODE.cpp:60-61
        flag("direct_orien_method", p.direct_orien_method); flag("multirate_ekf", p.multirate_ekf);
        flag("dynamic_meas_delay", p.dynamic_meas_delay);                                         // NODE.cpp:62-64
        vec("Q_a_diag", p.Q_a, 3, true); vec("Q_w_diag", p.Q_w, 3, true); vec("Q_ab_diag", p.Q_ab, 3, true); vec("Q_wb_diag", p.Q_wb, 3, true);
        vec("R_r_diag", p.R_r, 3, true); vec("R_ang_diag", p.R_ang, 3, true);                     // NODE.cpp:71-87
        num("r_cov_init", p.r_cov_init); num("v_cov_init", p.v_cov_init); num("ang_cov_init", p.ang_cov_init);
        num("ab_cov_init", p.ab_cov_init); num("wb_cov_init", p.wb_cov_init);                     // NODE.cpp:89-93
        vec("accel_bias_static", p.ab_static, 3, true); vec("gyro_bias_static", p.wb_static, 3, true);  // NODE.cpp:98-101
        vec("r_v_cv", p.r_v_cv, 3, true); vec("q_vc", p.q_vc, 4, true);                            // NODE.cpp:106-109
        double cw = p.camera_width, ch = p.camera_height, nt = p.n_tags;
        num("camera_width", cw); num("camera_height", ch); num("n_tags", nt);                      // NODE.cpp:112-113,119
        p.camera_width = (int32_t)std::lround(cw); p.camera_height = (int32_t)std::lround(ch); p.n_tags = (int32_t)std::lround(nt);
        vec("camera_K", p.camera_K, 9, true);                                                      // NODE.cpp:115-117
        num("tag_in_view_margin", p.tag_in_view_margin);                                           // NODE.cpp:120
        vec("tag_widths", p.tag_widths, QLE_MAX_TAGS, true); vec("tag_positions", p.tag_positions, 3 * QLE_MAX_TAGS, true);
    }
    if (update_freq > 0) p.update_freq = update_freq;
    if (measurement_freq > 0) p.measurement_freq = measurement_freq;
    if (corner_gate >= 0) p.corner_margin_enbl = corner_gate;
    if (multirate >= 0) p.multirate_ekf = multirate;
    qle_derived d;
    check(qle_params_derive(&p, &d), "qle_params_derive");
    if (!sequence.empty()) {
        int32_t nd = 0;
        check(qle_device_count(&nd), "qle_device_count");
        return run_recorded(p, d, sequence, trace, batch_given ? batch : 1, dtype, device);
    }

    // ---- shards: contiguous ranges of the global filter index, one per device
    int32_t ndev_avail = 0;
    check(qle_device_count(&ndev_avail), "qle_device_count");
    struct Shard {
        int dev = 0;
        int64_t lo = 0, n = 0;
        float ms = 0;
        double rm[3] = {0, 0, 0};
        int64_t bad = 0, tracked = 0;
        std::vector<double> pose, cov, vel, bias;
        std::vector<int32_t> upds;
        std::string err;
    };
    std::vector<Shard> shards((size_t)n_devices);
    {
        const int64_t base = batch / n_devices, extra = batch % n_devices;
        int64_t lo = 0;
        for (int k = 0; k < n_devices; ++k) {
            shards[(size_t)k].dev = (device + k) % std::max<int32_t>(ndev_avail, 1);
            shards[(size_t)k].lo = lo;
            shards[(size_t)k].n = base + (k < extra ? 1 : 0);
            lo += shards[(size_t)k].n;
        }
    }
    std::vector<uint8_t> has((size_t)ticks, 0);
    for (int64_t t = d.upd_per_meas - 1; t < ticks; t += d.upd_per_meas) has[(size_t)t] = 1;  // a tag pose every upd_per_meas ticks
    // every shard starts its timed ticks together: the threads meet here once their inputs are generated (a shard that failed on
    // the way still arrives, so nobody waits for it)
    std::atomic<int> arrived{0};
    auto meet = [&]() {
        arrived.fetch_add(1);
        while (arrived.load() < n_devices) std::this_thread::yield();
    };
    auto run_shard = [&](Shard& sh) {
        // errors are thread-local in the library: report them through the shard
        auto ok = [&](int rc, const char* what) {
            if (rc != QLE_OK && sh.err.empty()) sh.err = std::string(what) + ": " + qle_last_error();
            return rc == QLE_OK;
        };
        qle_batch* h = nullptr;
        qle_inputs* in = nullptr;
        bool met = false;
        if (sh.n == 0) { meet(); return; }
        do {
            if (!ok(qle_create(&h, sh.n, dtype, sh.dev, &p), "qle_create")) break;
            if (!ok(qle_enable_gating(h, 1), "qle_enable_gating")) break;
            if (!ok(qle_inputs_create(h, ticks, has.data(), &in), "qle_inputs_create")) break;
            qle_synth_cfg sc;
            qle_synth_cfg_default(&sc);
            sc.seed = seed;
            sc.filter_offset = sh.lo;  // data depend on the GLOBAL filter index only
            if (p.multirate_ekf) {
                // camera latency of the synthetic source = the configured measurement_delay (EKF.cpp:93,199)
                sc.meas_delay_ticks = d.measurement_step_delay;
                if (!ok(qle_set_uniform_measurement_age(h, p.dynamic_meas_delay ? d.measurement_step_delay * d.dT_nom - p.dyn_measurement_delay_offset
                                                                                   : p.measurement_delay), "qle_set_uniform_measurement_age")) break;
            }
            if (!ok(qle_synth_generate(h, in, &sc), "qle_synth_generate")) break;
            if (!ok(qle_synchronize(h), "qle_synchronize")) break;
            meet(); met = true;
            if (!ok(qle_timer_begin(h), "qle_timer_begin")) break;
            if (!ok(qle_run(h, in, 0, ticks), "qle_run")) break;
            if (!ok(qle_timer_end(h, &sh.ms), "qle_timer_end")) break;
            sh.pose.resize((size_t)sh.n * 7); sh.cov.resize((size_t)sh.n * 36); sh.vel.resize((size_t)sh.n * 3); sh.bias.resize((size_t)sh.n * 6);
            if (!ok(qle_get_report(h, sh.pose.data(), sh.cov.data(), sh.vel.data(), sh.bias.data()), "qle_get_report")) break;
            std::vector<uint8_t> perf((size_t)sh.n), cons((size_t)sh.n);
            sh.upds.resize((size_t)sh.n);
            if (!ok(qle_get_tick_flags(h, perf.data(), cons.data(), sh.upds.data()), "qle_get_tick_flags")) break;
            if (!ok(qle_synth_rmse(h, in, sh.rm), "qle_synth_rmse")) break;
            if (!ok(qle_count_nonfinite(h, &sh.bad), "qle_count_nonfinite")) break;
            for (int64_t i = 0; i < sh.n; ++i) sh.tracked += sh.upds[(size_t)i] < 2 * d.upd_per_meas ? 1 : 0;
        } while (false);
        if (!met) meet();
        qle_inputs_destroy(in);
        qle_destroy(h);
    };
    auto t0 = std::chrono::steady_clock::now();
    if (n_devices == 1) {
        run_shard(shards[0]);
    } else {
        std::vector<std::thread> th;
        for (auto& sh : shards) th.emplace_back(run_shard, std::ref(sh));
        for (auto& t : th) t.join();
    }
    double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (auto& sh : shards)
        if (!sh.err.empty()) { std::fprintf(stderr, "shard on device %d failed: %s\n", sh.dev, sh.err.c_str()); return 1; }
    // host-side combination: max of the device times, sums of the three RMSE scalars, filter 0's report
    float ms = 0;
    double rm[3] = {0, 0, 0};
    int64_t bad = 0, tracked = 0;
    for (auto& sh : shards) {
        ms = std::max(ms, sh.ms);
        for (int k = 0; k < 3; ++k) rm[k] += sh.rm[k];
        bad += sh.bad; tracked += sh.tracked;
    }
    const std::vector<double>&pose = shards[0].pose, &cov = shards[0].cov, &vel = shards[0].vel, &bias = shards[0].bias;
    const std::vector<int32_t>& upds = shards[0].upds;

    if (json) {
        // the fields of bench.py's line for the in-process sharding (one handle + stream + host thread per device, no torchrun): a fixed
        // population split over the devices (strong scaling), device time = max over the shards' HIP-event times of the same ticks
        std::printf("{\"metric\": \"EKF predict+update steps/sec; in-process sharding over devices\", \"value\": %.6e, \"unit\": \"EKF ticks/s\", "
                    "\"n_gpus\": %d, \"devices_present\": %d, \"steps\": %lld, \"warmup\": 0, \"ms_per_step\": %.6f, \"higher_is_better\": true, "
                    "\"scaling\": \"strong\", \"vs_baseline\": null, \"dtype\": \"%s\", \"data\": \"synthetic\", "
                    "\"config\": {\"workload\": \"ekf_driver: %lld filters x %lld ticks, tag pose every %d ticks, multirate %d\", \"global_batch\": %lld, "
                    "\"parallelism\": \"filters sharded x%d in one process, no collectives\"}, \"per_shard\": [",
                    (double)batch * ticks / (ms * 1e-3), n_devices, (int)ndev_avail, (long long)ticks, ms / (double)ticks, dtype == QLE_F32 ? "f32" : "f64",
                    (long long)batch, (long long)ticks, d.upd_per_meas, p.multirate_ekf, (long long)batch, n_devices);
        for (size_t k = 0; k < shards.size(); ++k)
            std::printf("%s{\"device\": %d, \"filter_offset\": %lld, \"filters\": %lld, \"hip_event_ms\": %.4f}", k ? ", " : "", shards[k].dev,
                        (long long)shards[k].lo, (long long)shards[k].n, shards[k].ms);
        std::printf("], \"wall_ms_incl_setup\": %.3f, \"rmse_vs_truth\": {\"position_m\": %.6f, \"attitude_rad\": %.6f, \"filters\": %.0f}, "
                    "\"nonfinite_filters\": %lld}\n",
                    wall * 1e3, std::sqrt(rm[0] / rm[2]), std::sqrt(rm[1] / rm[2]), rm[2], (long long)bad);
        return bad == 0 ? 0 : 3;
    }
    std::printf("%s\n", qle_version());
    std::printf("params: update_freq %.1f Hz, measurement_freq %.1f Hz (every %d ticks), num_states %d, direct_orien_method %d, limit %d, corner gate %d, n_tags %d, multirate %d (step delay %d)\n",
                p.update_freq, p.measurement_freq, d.upd_per_meas, d.num_states, p.direct_orien_method, p.limit_measurement_freq, p.corner_margin_enbl, p.n_tags,
                p.multirate_ekf, d.measurement_step_delay);
    std::printf("batch %lld x %lld ticks (%s) on %d device shard(s) [%d device(s) present]: %.3f ms device (max over shards), %.3f ms wall incl. setup -> %.3e ticks/s\n",
                (long long)batch, (long long)ticks, dtype == QLE_F32 ? "fp32" : "fp64", n_devices, (int)ndev_avail, ms, wall * 1e3,
                (double)batch * ticks / (ms * 1e-3));
    std::printf("filter 0: rel_pose position (%.4f, %.4f, %.4f) orientation xyzw (%.4f, %.4f, %.4f, %.4f)\n", pose[0], pose[1], pose[2], pose[3], pose[4], pose[5], pose[6]);
    std::printf("filter 0: velocity (%.4f, %.4f, %.4f)  IMU bias+static accel (%.4f, %.4f, %.4f) gyro (%.5f, %.5f, %.5f)\n", vel[0], vel[1], vel[2], bias[0], bias[1],
                bias[2], bias[3], bias[4], bias[5]);
    std::printf("filter 0: pose covariance diag (%.3e, %.3e, %.3e, %.3e, %.3e, %.3e)  upds_since_correction %d\n", cov[0], cov[7], cov[14], cov[21], cov[28], cov[35], upds[0]);
    std::printf("filters corrected within the last %d ticks: %lld of %lld (the corner gate, EKF.cpp:156-186, rejects tags outside the image margins)\n",
                2 * d.upd_per_meas, (long long)tracked, (long long)batch);
    std::printf("RMSE vs synthetic truth over %.0f filters: position %.4f m, attitude %.4f rad; non-finite filters: %lld\n", rm[2], std::sqrt(rm[0] / rm[2]),
                std::sqrt(rm[1] / rm[2]), (long long)bad);
    return bad == 0 ? 0 : 3;
}
