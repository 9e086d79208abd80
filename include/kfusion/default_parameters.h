// kfusion/default_parameters.h — command line of the SLAMBench KinectFusion drivers.
//
// Restates the reference's `struct Configuration` (kfusion/include/default_parameters.h:25-43 defaults,
// :53-75 option table, :155-230 comma-list parsing, :232-420 validation): same short and long options, same
// defaults, same accepted ranges, same exit codes.  `-v/-s/-p` take one or three comma-separated values (one
// value is broadcast), `-k` takes fx,fy,cx,cy, `-y` lists ICP iterations per pyramid level, finest level first.
#ifndef KFUSION_HIP_DEFAULT_PARAMETERS_H_
#define KFUSION_HIP_DEFAULT_PARAMETERS_H_

#include <getopt.h>
#include <sys/stat.h>

#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "commons.h"

// kfusion/include/constant_parameters.h:15-26 (compile-time constants of the algorithm)
const float e_delta = 0.1f;
const int radius = 2;
const float dist_threshold = 0.1f;
const float normal_threshold = 0.8f;
const float track_threshold = 0.15f;
const float maxweight = 100.0f;
const float nearPlane = 0.4f;
const float farPlane = 4.0f;
const float delta = 4.0f;

// run-time defaults (default_parameters.h:25-43)
const float default_mu = 0.1f;
const bool default_blocking_read = false;
const int default_fps = 0;
const float default_icp_threshold = 1e-5;
const int default_compute_size_ratio = 1;
const int default_integration_rate = 2;
const int default_rendering_rate = 4;
const int default_tracking_rate = 1;
const uint3 default_volume_resolution = {256, 256, 256};
const float3 default_volume_size = {2.f, 2.f, 2.f};
const float3 default_initial_pos_factor = {0.5f, 0.5f, 0.0f};
static const int default_iterations[3] = {10, 5, 4};

struct Configuration {
    int compute_size_ratio, integration_rate, rendering_rate, tracking_rate;
    uint3 volume_resolution;
    float3 volume_size, initial_pos_factor;
    std::vector<int> pyramid;
    std::string dump_volume_file, input_file, log_file;
    std::ofstream log_filestream;      // default_parameters.h:92-93: the GUI mains open the log themselves
    std::ostream *log_stream;
    float4 camera;
    bool camera_overrided;
    float mu;
    int fps;
    bool blocking_read;
    float icp_threshold;
    bool no_gui, render_volume_fullsize;

    static std::vector<std::string> split(const char *arg) {
        std::vector<std::string> out;
        std::istringstream in(arg);
        std::string item;
        while (std::getline(in, item, ',')) out.push_back(item);
        return out;
    }
    // one value is broadcast to the rest, two leave the third equal to the second (default_parameters.h:155-230)
    static void spread(const std::vector<std::string> &items, double *dst, int n) {
        for (int i = 0; i < n; ++i) {
            if (i < (int)items.size()) dst[i] = atof(items[i].c_str());
            else if (i > 0) dst[i] = dst[i - 1];
        }
    }
    static float3 atof3(const char *arg) { double v[3] = {0, 0, 0}; spread(split(arg), v, 3); return make_float3(v[0], v[1], v[2]); }
    static uint3 atoi3(const char *arg) {
        const std::vector<std::string> it = split(arg);
        int v[3] = {0, 0, 0};
        for (int i = 0; i < 3; ++i) v[i] = i < (int)it.size() ? atoi(it[i].c_str()) : (i ? v[i - 1] : 0);
        return make_uint3(v[0], v[1], v[2]);
    }
    static float4 atof4(const char *arg) { double v[4] = {0, 0, 0, 0}; spread(split(arg), v, 4); return make_float4(v[0], v[1], v[2], v[3]); }

    void print_arguments() {
        std::cerr << "-c  (--compute-size-ratio)       : default is " << default_compute_size_ratio << "   (same size)      \n"
                  << "-d  (--dump-volume) <filename>   : Output volume file              \n"
                  << "-f  (--fps)                      : default is " << default_fps << "\n"
                  << "-i  (--input-file) <filename>    : Input camera file               \n"
                  << "-k  (--camera)                   : default is defined by input     \n"
                  << "-l  (--icp-threshold)                : default is " << default_icp_threshold << "\n"
                  << "-o  (--log-file) <filename>      : default is stdout               \n"
                  << "-m  (--mu)                       : default is " << default_mu << "               \n"
                  << "-p  (--init-pose)                : default is " << default_initial_pos_factor.x << "," << default_initial_pos_factor.y << "," << default_initial_pos_factor.z << "     \n"
                  << "-q  (--no-gui)                   : default is to display gui\n"
                  << "-r  (--integration-rate)         : default is " << default_integration_rate << "     \n"
                  << "-s  (--volume-size)              : default is " << default_volume_size.x << "," << default_volume_size.y << "," << default_volume_size.z << "      \n"
                  << "-t  (--tracking-rate)            : default is " << default_tracking_rate << "     \n"
                  << "-v  (--volume-resolution)        : default is " << default_volume_resolution.x << "," << default_volume_resolution.y << "," << default_volume_resolution.z << "    \n"
                  << "-y  (--pyramid-levels)           : default is 10,5,4     \n"
                  << "-z  (--rendering-rate)   : default is " << default_rendering_rate << std::endl;
    }

    // report header of the GUI mains' log (default_parameters.h:124-154); the first line carries the wall clock
    void print_values(std::ostream &out) {
        char stamp[80];
        const time_t now = time(NULL);
        strftime(stamp, sizeof(stamp), "%Y-%m-%d %I:%M:%S", localtime(&now));
        out << "SLAMBench Report run started:\t" << stamp << "\n\n"
            << "Scene properties:\n=================\n\n"
            << "input-file: " << input_file << "\n"
            << "volume-size: " << volume_size.x << "," << volume_size.y << "," << volume_size.z << "\n"
            << "camera: " << camera.x << "," << camera.y << "," << camera.z << "," << camera.w << "\n"
            << "init-pose: " << initial_pos_factor.x << "," << initial_pos_factor.y << "," << initial_pos_factor.z << "\n\n"
            << "Algorithmic properties:\n=======================\n\n"
            << "compute-size-ratio: " << compute_size_ratio << "\n"
            << "volume-resolution: " << volume_resolution.x << "," << volume_resolution.y << "," << volume_resolution.z << "    \n"
            << "mu: " << mu << "\n"
            << "icp-threshold: " << icp_threshold << "\n"
            << "pyramid-levels: ";
        for (size_t i = 0; i < pyramid.size(); ++i) out << (i ? "," : "") << pyramid[i];
        out << "\ntracking-rate: " << tracking_rate << "\nintegration-rate: " << integration_rate << "\nrendering-rate: " << rendering_rate
            << "\nfps: " << fps << std::endl;
    }

    Configuration(unsigned int argc, char **argv)
        : compute_size_ratio(default_compute_size_ratio), integration_rate(default_integration_rate),
          rendering_rate(default_rendering_rate), tracking_rate(default_tracking_rate),
          volume_resolution(default_volume_resolution), volume_size(default_volume_size),
          initial_pos_factor(default_initial_pos_factor), pyramid(default_iterations, default_iterations + 3), log_stream(NULL),
          camera_overrided(false), mu(default_mu), fps(default_fps), blocking_read(default_blocking_read),
          icp_threshold(default_icp_threshold), no_gui(false), render_volume_fullsize(false) {
        camera = make_float4(0, 0, 0, 0);
        static const struct option long_options[] = {
            {"compute-size-ratio", required_argument, 0, 'c'}, {"dump-volume", required_argument, 0, 'd'},
            {"fps", required_argument, 0, 'f'},                {"input-file", required_argument, 0, 'i'},
            {"camera", required_argument, 0, 'k'},             {"icp-threshold", required_argument, 0, 'l'},
            {"log-file", required_argument, 0, 'o'},           {"mu", required_argument, 0, 'm'},
            {"init-pose", required_argument, 0, 'p'},          {"no-gui", no_argument, 0, 'q'},
            {"integration-rate", required_argument, 0, 'r'},   {"volume-size", required_argument, 0, 's'},
            {"tracking-rate", required_argument, 0, 't'},      {"volume-resolution", required_argument, 0, 'v'},
            {"pyramid-levels", required_argument, 0, 'y'},     {"rendering-rate", required_argument, 0, 'z'},
            {0, 0, 0, 0}};
        int errors = 0, c, index = 0;
        optind = 1;
        while ((c = getopt_long(argc, argv, "qc:d:f:i:l:m:k:o:p:r:s:t:v:y:z:", long_options, &index)) != -1) {
            switch (c) {
            case 'c':
                compute_size_ratio = atoi(optarg);
                std::cerr << "update compute_size_ratio to " << compute_size_ratio << std::endl;
                if (compute_size_ratio != 1 && compute_size_ratio != 2 && compute_size_ratio != 4 && compute_size_ratio != 8) {
                    std::cerr << "ERROR: --compute-size-ratio (-c) must be 1, 2 ,4 or 8  (was " << optarg << ")\n";
                    ++errors;
                }
                break;
            case 'd': dump_volume_file = optarg; std::cerr << "update dump_volume_file to " << dump_volume_file << std::endl; break;
            case 'f':
                fps = atoi(optarg);
                std::cerr << "update fps to " << fps << std::endl;
                if (fps < 0) { std::cerr << "ERROR: --fps (-f) must be >= 0 (was " << optarg << ")\n"; ++errors; }
                break;
            case 'i': {
                input_file = optarg;
                std::cerr << "update input_file to " << input_file << std::endl;
                struct stat st;
                if (stat(input_file.c_str(), &st) != 0) {
                    std::cerr << "ERROR: --input-file (-i) does not exist (was " << input_file << ")\n";
                    ++errors;
                }
                break;
            }
            case 'k':
                camera = atof4(optarg);
                camera_overrided = true;
                std::cerr << "update camera to " << camera.x << "," << camera.y << "," << camera.z << "," << camera.w << std::endl;
                break;
            case 'o': log_file = optarg; std::cerr << "update log_file to " << log_file << std::endl; break;
            case 'l': icp_threshold = atof(optarg); std::cerr << "update icp_threshold to " << icp_threshold << std::endl; break;
            case 'm': mu = atof(optarg); std::cerr << "update mu to " << mu << std::endl; break;
            case 'p':
                initial_pos_factor = atof3(optarg);
                std::cerr << "update init_poseFactors to " << initial_pos_factor.x << "," << initial_pos_factor.y << "," << initial_pos_factor.z << std::endl;
                break;
            case 'q': no_gui = true; break;
            case 'r':
                integration_rate = atoi(optarg);
                std::cerr << "update integration_rate to " << integration_rate << std::endl;
                if (integration_rate < 1) { std::cerr << "ERROR: --integration-rate (-r) must >= 1 (was " << optarg << ")\n"; ++errors; }
                break;
            case 's':
                volume_size = atof3(optarg);
                std::cerr << "update map_size to " << volume_size.x << "mx" << volume_size.y << "mx" << volume_size.z << "m" << std::endl;
                if (volume_size.x <= 0 || volume_size.y <= 0 || volume_size.z <= 0) {
                    std::cerr << "ERROR: --volume-size (-s) all dimensions must > 0 (was " << optarg << ")\n";
                    ++errors;
                }
                break;
            case 't': tracking_rate = atof(optarg); std::cerr << "update tracking_rate to " << tracking_rate << std::endl; break;
            case 'z': rendering_rate = atof(optarg); std::cerr << "update rendering_rate to " << rendering_rate << std::endl; break;
            case 'v':
                volume_resolution = atoi3(optarg);
                std::cerr << "update volumetric_size to " << volume_resolution.x << "x" << volume_resolution.y << "x" << volume_resolution.z << std::endl;
                if (volume_resolution.x <= 0 || volume_resolution.y <= 0 || volume_resolution.z <= 0) {
                    std::cerr << "ERROR: --volume-size (-s) all dimensions must > 0 (was " << optarg << ")\n";
                    ++errors;
                }
                break;
            case 'y': {
                pyramid.clear();
                const std::vector<std::string> levels = split(optarg);
                for (size_t i = 0; i < levels.size(); ++i) pyramid.push_back((int)atof(levels[i].c_str()));
                std::cerr << "update pyramid levels to ";
                for (size_t i = 0; i < pyramid.size(); ++i) std::cerr << pyramid[i] << " ";
                std::cerr << std::endl;
                break;
            }
            case 0:
            case '?':
                std::cerr << "Unknown option character -" << char(optopt) << " or bad usage.\n";
                print_arguments();
                exit(0);
            default:
                std::cerr << "GetOpt abort.";
                errors = 1;
            }
        }
        if (errors) {
            std::cerr << "Exited due to " << errors << " error" << (errors == 1 ? "" : "s") << " in command line options\n";
            exit(1);
        }
    }
};

#endif
