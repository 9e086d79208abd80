// kfusion-benchmark-hip — the SLAMBench "benchmark" application mode on the MI355X HIP backend.
//
// Same command line, same per-frame call sequence and the same tab-separated log as the reference driver
// (kfusion/src/benchmark.cpp:47-190): preprocessing, tracking, integration, raycasting and the three renders
// are timed with a device synchronisation at each boundary (benchmark.cpp:26-39), one row per frame:
//   frame acquisition preprocessing tracking integration raycasting rendering computation total X Y Z tracked integrated
// so the reference's checkPos.py-style tooling (slambench_amd/checkpos.py) reads it unchanged.
#include <cassert>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>

#include <kfusion/default_parameters.h>
#include <kfusion/interface.h>
#include <kfusion/kernels.h>

static double tock() {
    synchroniseDevices();
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + t.tv_nsec / 1000000000.0;
}

int main(int argc, char **argv) {
    Configuration config(argc, argv);
    assert(config.compute_size_ratio > 0);
    assert(config.integration_rate > 0);
    assert(config.volume_size.x > 0);
    assert(config.volume_resolution.x > 0);

    std::ostream *log = &std::cout;
    std::ofstream logfile;
    if (config.log_file != "") {
        logfile.open(config.log_file.c_str());
        log = &logfile;
    }
    if (config.input_file == "") {
        std::cerr << "No input found." << std::endl;
        config.print_arguments();
        return 1;
    }

    DepthReader *reader = is_file(config.input_file)
                              ? static_cast<DepthReader *>(new RawDepthReader(config.input_file, config.fps, config.blocking_read))
                              : static_cast<DepthReader *>(new SceneDepthReader(config.input_file, config.fps, config.blocking_read));
    std::cout.precision(10);
    std::cerr.precision(10);

    const float3 init_pose = config.initial_pos_factor * config.volume_size;
    const uint2 inputSize = reader->getinputSize();
    std::cerr << "input Size is = " << inputSize.x << "," << inputSize.y << std::endl;
    if (!reader->isValid() || inputSize.x == 0 || inputSize.y == 0) return 1;

    const uint2 computationSize = make_uint2(inputSize.x / config.compute_size_ratio, inputSize.y / config.compute_size_ratio);
    float4 camera = reader->getK() / (float)config.compute_size_ratio;
    if (config.camera_overrided) camera = config.camera / (float)config.compute_size_ratio;

    const size_t in_px = (size_t)inputSize.x * inputSize.y, out_px = (size_t)computationSize.x * computationSize.y;
    unsigned short *inputDepth = (unsigned short *)malloc(sizeof(unsigned short) * in_px);
    uchar4 *depthRender = (uchar4 *)malloc(sizeof(uchar4) * out_px);
    uchar4 *trackRender = (uchar4 *)malloc(sizeof(uchar4) * out_px);
    uchar4 *volumeRender = (uchar4 *)malloc(sizeof(uchar4) * out_px);

    init();
    {
        Kfusion kfusion(computationSize, config.volume_resolution, config.volume_size, init_pose, config.pyramid);

        double t[7];
        t[0] = tock();
        *log << "frame\tacquisition\tpreprocessing\ttracking\tintegration\traycasting\trendering\tcomputation\ttotal    \tX          \tY          \tZ         \ttracked   \tintegrated"
             << std::endl;
        log->setf(std::ios::fixed, std::ios::floatfield);

        unsigned int frame = 0;
        while (reader->readNextDepthFrame(inputDepth)) {
            const Matrix4 pose = kfusion.getPose();   // the pose *before* this frame is what gets logged
            const float xt = pose.data[0].w - init_pose.x, yt = pose.data[1].w - init_pose.y, zt = pose.data[2].w - init_pose.z;
            t[1] = tock();
            kfusion.preprocessing(inputDepth, inputSize);
            t[2] = tock();
            const bool tracked = kfusion.tracking(camera, config.icp_threshold, config.tracking_rate, frame);
            t[3] = tock();
            const bool integrated = kfusion.integration(camera, config.integration_rate, config.mu, frame);
            t[4] = tock();
            kfusion.raycasting(camera, config.mu, frame);
            t[5] = tock();
            kfusion.renderDepth(depthRender, computationSize);
            kfusion.renderTrack(trackRender, computationSize);
            kfusion.renderVolume(volumeRender, computationSize, frame, config.rendering_rate, camera, 0.75 * config.mu);
            t[6] = tock();

            *log << frame << "\t" << t[1] - t[0] << "\t" << t[2] - t[1] << "\t" << t[3] - t[2] << "\t" << t[4] - t[3] << "\t"
                 << t[5] - t[4] << "\t" << t[6] - t[5] << "\t" << t[5] - t[1] << "\t" << t[6] - t[0] << "\t" << xt << "\t" << yt
                 << "\t" << zt << "\t" << tracked << "        \t" << integrated << std::endl;
            ++frame;
            t[0] = tock();
        }
        if (config.dump_volume_file != "") kfusion.dumpVolume(config.dump_volume_file.c_str());
    }
    clean();
    free(inputDepth);
    free(depthRender);
    free(trackRender);
    free(volumeRender);
    delete reader;
    return 0;
}
