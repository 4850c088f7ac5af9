// eval_tool.cpp — classification harness, counterpart of the reference's eval_tool
// (eval_tool/eval_classification.cpp:43-585): same flags (-t / -d <ism>, -f <list>, -o <folder>, -i), same list format
// (eval_helpers.h:100-177), predicted class = maxima[0].classId (-1 if none), accuracy + mean per-class accuracy, the seven
// timer keys and the summary.txt layout (:412-558). New: -b <objects per batch> feeds detectBatch() so that descriptors,
// kNN, votes and maxima of many objects stay on the MI355X between stages.
#include <sys/stat.h>

#include <chrono>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "ism3d.h"

using namespace ism3d;

static void usage() {
    std::cout << "Generic options:\n  -h [ --help ]            Display this help message\n"
                 "  -o [ --output ] arg      The output folder for ism files after training or the classification log\n"
                 "  -f [ --inputfile ] arg   Input file (for training or testing) containing the input clouds and their labels\n"
                 "  -b [ --batch ] arg       objects per device batch in detection (default 32)\n"
                 "Training:\n  -t [ --train ] arg       Train an implicit shape model\n  -i [ --inplace ]         Overwrite the loaded ism file\n"
                 "Detection:\n  -d [ --detect ] arg      Detect using a trained implicit shape model\n";
}

int main(int argc, char** argv) {
    std::string out_dir, list_file, train_ism, detect_ism;
    bool inplace = false;
    size_t batch = 32;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << "missing value for " << a << std::endl; exit(1); } return argv[++i]; };
        if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (a == "-o" || a == "--output") out_dir = next();
        else if (a == "-f" || a == "--inputfile") list_file = next();
        else if (a == "-t" || a == "--train") train_ism = next();
        else if (a == "-d" || a == "--detect") detect_ism = next();
        else if (a == "-b" || a == "--batch") batch = (size_t)std::max(1, atoi(next().c_str()));
        else if (a == "-i" || a == "--inplace") inplace = true;
        else { std::cerr << "unknown option " << a << std::endl; usage(); return 1; }
    }
    if (list_file.empty() || (train_ism.empty() && detect_ism.empty())) { usage(); return 1; }
    try {
        FileList L = parseFileList(list_file);
        ImplicitShapeModel ism;
        ism.setLogging(false);
        ism.setSignalsState(false);
        if (!train_ism.empty()) {
            if (L.mode != "train") { std::cerr << "ERROR: Check your command line arguments! You specified to train, but your input file says test!" << std::endl; return 1; }
            if (!ism.readObject(train_ism, true)) { std::cerr << "could not read ism from file, training stopped: " << train_ism << std::endl; return 1; }
            for (size_t i = 0; i < L.filenames.size(); ++i)
                if (!ism.addTrainingModel(L.filenames[i], L.class_labels[i], L.instance_labels[i])) return 1;
            ism.train();
            ism.setLabels(L.class_labels_rmap, L.instance_labels_rmap, L.instance_to_class_map);
            std::string out_file = train_ism;
            if (!inplace) {
                if (out_dir.empty()) { std::cerr << "no output file specified" << std::endl; return 1; }
                mkdir(out_dir.c_str(), 0755);
                const size_t p = train_ism.find_last_of('/');
                out_file = out_dir + "/" + (p == std::string::npos ? train_ism : train_ism.substr(p + 1));
            }
            if (!ism.writeObject(out_file)) return 1;
            std::cout << "trained model written to " << out_file << " (" << ism.getCodebook()->getSize() << " codewords)" << std::endl;
            return 0;
        }
        if (L.mode != "test") { std::cerr << "ERROR: Check your command line arguments! You specified to detect, but your input file says train!" << std::endl; return 1; }
        if (!ism.readObject(detect_ism)) { std::cerr << "could not read ism from file, detection stopped: " << detect_ism << std::endl; return 1; }
        if (out_dir.empty()) out_dir = ".";
        mkdir(out_dir.c_str(), 0755);
        std::ofstream summary(out_dir + "/summary.txt");
        const auto t_start = std::chrono::steady_clock::now();
        unsigned numCorrectClasses = 0, numCorrectInstances = 0;
        std::map<unsigned, std::pair<unsigned, unsigned>> perClass;
        std::map<std::string, double> times;
        for (size_t b = 0; b < L.filenames.size(); b += batch) {
            const size_t e = std::min(L.filenames.size(), b + batch);
            std::vector<std::shared_ptr<PointCloud>> owned;
            std::vector<const PointCloud*> ptrs;
            for (size_t i = b; i < e; ++i) {
                auto c = ImplicitShapeModel::loadPointCloud(L.filenames[i]);
                if (!c) { std::cerr << "detection failed: " << L.filenames[i] << std::endl; return 1; }
                owned.push_back(c); ptrs.push_back(c.get());
            }
            auto res = ism.detectBatch(ptrs);
            times = ism.getProcessingTimes();
            for (size_t i = b; i < e; ++i) {
                const auto& maxima = res[i - b];
                int classId = -1, instanceId = -1;
                if (!maxima.empty()) { classId = (int)maxima[0].classId; instanceId = (int)maxima[0].instanceId; }
                const unsigned trueClassID = L.class_labels[i], trueInstanceID = L.instance_labels[i];
                summary << "file: " << L.filenames[i] << ", ground truth class: " << trueClassID << ", classified class: " << classId << std::endl;
                auto& pc = perClass[trueClassID];
                pc.second++;
                if ((int)trueClassID == classId) { numCorrectClasses++; pc.first++; }
                if ((int)trueInstanceID == instanceId) numCorrectInstances++;
            }
        }
        const size_t n = L.filenames.size();
        summary << "\n\nclass id to class name mapping:" << std::endl;
        for (auto& el : L.class_labels_rmap) summary << el.first << ": " << el.second << std::endl;
        double time_sum = 0;
        for (auto& it : times) if (it.first != "complete") time_sum += it.second / 1000;
        summary << "\n\n\ncomplete time: " << times["complete"] / 1000 << " [s]" << ", sum all steps: " << time_sum << " [s]" << std::endl;
        summary << "times per step:\n";
        summary << "create flann index: " << std::setw(10) << std::setfill(' ') << times["flann"] / 1000 << " [s]" << std::endl;
        summary << "compute normals:    " << std::setw(10) << std::setfill(' ') << times["normals"] / 1000 << " [s]" << std::endl;
        summary << "compute keypoints:  " << std::setw(10) << std::setfill(' ') << times["keypoints"] / 1000 << " [s]" << std::endl;
        summary << "compute features:   " << std::setw(10) << std::setfill(' ') << times["features"] / 1000 << " [s]" << std::endl;
        summary << "cast votes:         " << std::setw(10) << std::setfill(' ') << times["voting"] / 1000 << " [s]" << std::endl;
        summary << "find maxima:        " << std::setw(10) << std::setfill(' ') << times["maxima"] / 1000 << " [s]" << std::endl;
        float avg_pc_acc = 0;
        for (auto& el : perClass) avg_pc_acc += (float)el.second.first / el.second.second;
        avg_pc_acc /= perClass.size();
        summary << std::endl << std::endl;
        summary << " Accuracy: " << ((float)numCorrectClasses / n) * 100.0f << " %, Average per Class Accuracy: " << avg_pc_acc * 100.0f << " %" << std::endl << std::endl;
        summary << " result: " << numCorrectClasses << " of " << n << " clouds classified correctly (" << ((float)numCorrectClasses / n) * 100.0f << " %)\n";
        summary << " result: " << numCorrectInstances << " of " << n << " instances recognized correctly (" << ((float)numCorrectInstances / n) * 100.0f << " %)\n\n";
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        summary << " Total processing time: " << std::fixed << std::setprecision(4) << wall << " seconds \n";
        std::cout << "Accuracy: " << ((float)numCorrectClasses / n) * 100.0f << " % (" << numCorrectClasses << " of " << n << "), summary in " << out_dir << "/summary.txt" << std::endl;
        return 0;
    } catch (const ism3d::Exception& e) {       // eval_classification.cpp:574-581
        std::cerr << e.what() << std::endl;
        return 1;
    }
}
