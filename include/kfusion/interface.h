// kfusion/interface.h — depth stream readers for the SLAMBench drivers.
//
// DepthReader / RawDepthReader / SceneDepthReader with the reference's interface
// (kfusion/include/interface.h:28-87, :199-318, :99-197).  The `.raw` wire format per frame is
//   uint32 W, H | W*H uint16 depth (mm) | uint32 W, H | W*H*3 uint8 rgb        (interface.h:244-283)
// and an ICL-NUIM scene directory holds scene_00_%04d.depth text files of *ray lengths* in metres, which are
// converted to z-depth with the ICL-NUIM intrinsics (interface.h:157-195).  OpenNI readers are out of scope.
#ifndef KFUSION_HIP_INTERFACE_H_
#define KFUSION_HIP_INTERFACE_H_

#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "commons.h"

enum ReaderType { READER_RAW, READER_SCENE, READER_OPENNI };

class DepthReader {
public:
    virtual ~DepthReader() {}
    virtual bool readNextDepthFrame(float *depthMap) = 0;
    inline bool readNextDepthFrame(unsigned short int *UintdepthMap) { return readNextDepthFrame(NULL, UintdepthMap); }
    virtual bool readNextDepthFrame(uchar3 *raw_rgb, unsigned short int *depthMap) = 0;
    virtual float4 getK() = 0;
    virtual uint2 getinputSize() = 0;
    virtual void restart() = 0;
    virtual ReaderType getType() = 0;
    bool isValid() { return cameraOpen; }
    int getFrameNumber() { return _frame; }

    // frame pacing (interface.h:48-82): fps == 0 processes every frame; otherwise frames are dropped to follow
    // the wall clock, and --blocking-read sleeps until the next frame is due
    void get_next_frame() {
        if (_fps == 0) { ++_frame; return; }
        struct timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        const double t = now.tv_sec + now.tv_nsec / 1000000000.0;
        static double first = t;
        _frame = (int)std::ceil((t - first) * (double)_fps);
        std::cout << "next frame is " << _frame << std::endl;
        const double wait = (double)_frame / _fps - t + first;
        if (_blocking_read && wait > 0) usleep((useconds_t)(1000000.0 * wait));
    }

    bool cameraActive, cameraOpen;

protected:
    int _frame, _fps;
    bool _blocking_read;
};

class RawDepthReader : public DepthReader {
    FILE *file_;
    uint2 size_;
    unsigned int hdr_[2];   // header of the frame read last: the reference's reader keeps it in an uninitialised local, so at the
                            // end of the file the stale value decides between "End of file." and "End of file(garbage found)."
    std::vector<unsigned short> scratch_;

public:
    RawDepthReader(std::string filename, int fps, bool blocking_read) : file_(fopen(filename.c_str(), "rb")) {
        cameraOpen = cameraActive = false;
        size_ = make_uint2(0, 0);
        hdr_[0] = hdr_[1] = 0;
        _frame = -1; _fps = fps; _blocking_read = blocking_read;
        unsigned int hdr[2];
        if (!file_ || fread(hdr, sizeof hdr, 1, file_) != 1) {
            std::cerr << "Invalid Raw file." << std::endl;
            return;
        }
        size_ = make_uint2(hdr[0], hdr[1]);
        cameraOpen = cameraActive = true;
        fseek(file_, 0, SEEK_SET);
        scratch_.resize((size_t)size_.x * size_.y);
    }
    ~RawDepthReader() { if (file_) fclose(file_); }
    ReaderType getType() { return READER_RAW; }
    uint2 getinputSize() { return size_; }
    float4 getK() { return make_float4(531.15f, 531.15f, 640 / 2, 480 / 2); }   // interface.h:313-315
    void restart() { _frame = -1; if (file_) rewind(file_); }

    bool readNextDepthFrame(uchar3 *raw_rgb, unsigned short int *depthMap) {
        if (!file_) return false;
        get_next_frame();
        const size_t pixels = (size_t)size_.x * size_.y;
        const size_t frame_bytes = 4 * sizeof(unsigned int) + pixels * sizeof(unsigned short) + pixels * 3;
        if (fseek(file_, (long)(frame_bytes * (size_t)_frame), SEEK_SET) != 0) return false;
        size_t got = 0, want = 0;
        unsigned int *hdr = hdr_;
        got += fread(hdr, sizeof hdr_, 1, file_);
        const size_t n = (size_t)hdr[0] * hdr[1];
        want += 1 + n;
        if (depthMap) got += fread(depthMap, sizeof(unsigned short), n, file_);
        else { got += n; fseek(file_, (long)(n * sizeof(unsigned short)), SEEK_CUR); }
        got += fread(hdr, sizeof hdr_, 1, file_);
        const size_t m = (size_t)hdr[0] * hdr[1];
        want += 1 + m;
        if (raw_rgb) got += fread(raw_rgb, 3, m, file_);
        else { got += m; fseek(file_, (long)(m * 3), SEEK_CUR); }
        if (got != want) {
            std::cout << "End of file" << (got == 0 ? "" : "(garbage found)") << "." << std::endl;
            return false;
        }
        return true;
    }
    bool readNextDepthFrame(float *depthMap) {
        const bool ok = readNextDepthFrame(NULL, scratch_.data());
        for (size_t i = 0; i < scratch_.size(); ++i) depthMap[i] = (float)scratch_[i] / 1000.0f;
        return ok;
    }
};

// ICL-NUIM intrinsics of the scene files (interface.h:89-97)
static const float kSceneFx = 481.20f, kSceneFy = -480.00f, kSceneU0 = 319.50f, kSceneV0 = 239.50f;

class SceneDepthReader : public DepthReader {
    std::string dir_;
    uint2 size_;
    std::vector<float> scratch_;

public:
    SceneDepthReader(std::string dir, int fps, bool blocking_read) : dir_(dir), size_(make_uint2(640, 480)) {
        struct stat st;
        cameraOpen = cameraActive = (lstat(dir.c_str(), &st) == 0 && S_ISDIR(st.st_mode));
        if (!cameraOpen) std::cerr << "No such directory " << dir << std::endl;
        _frame = -1; _fps = fps; _blocking_read = blocking_read;
        scratch_.resize((size_t)size_.x * size_.y);
    }
    ReaderType getType() { return READER_SCENE; }
    float4 getK() { return make_float4(481.20f, 480.00f, 319.50f, 239.50f); }
    uint2 getinputSize() { return size_; }
    void restart() { _frame = 0; }

    bool readNextDepthFrame(uchar3 *, unsigned short int *depthMap) {
        const bool ok = readNextDepthFrame(scratch_.data());
        for (size_t i = 0; i < scratch_.size(); ++i) depthMap[i] = (unsigned short)(scratch_[i] * 1000.0f);
        return ok;
    }
    bool readNextDepthFrame(float *depthMap) {
        get_next_frame();
        std::ostringstream name;
        name << dir_ << "/scene_00_" << std::setfill('0') << std::setw(4) << _frame << ".depth";
        std::ifstream in(name.str().c_str());
        if (!in) {
            std::cerr << "Can't open Data from " << name.str() << "!\n";
            return false;
        }
        const size_t pixels = (size_t)size_.x * size_.y;
        size_t count = 0;
        float d;
        while (count < pixels && (in >> d)) depthMap[count++] = d;
        // ray length -> z-depth (interface.h:181-193)
        for (unsigned v = 0; v < size_.y; ++v)
            for (unsigned u = 0; u < size_.x; ++u) {
                const float a = (u - kSceneU0) / kSceneFx, b = (v - kSceneV0) / kSceneFy;
                depthMap[u + v * size_.x] = depthMap[u + v * size_.x] / std::sqrt(a * a + b * b + 1);
            }
        return count > 0;
    }
};

#endif
