// spu_compat.hpp -- the SUBSET of the StreamPU surface the DVB-S2 RX inner path touches
// (SURVEY.md 8b): Module / Task / Socket / codelet, socket binding, a run-in-order Sequence and
// the spu::tools exceptions.  lib/streampu is an empty submodule in the reference snapshot, so
// this header exists only so that the module headers in this directory compile and run
// stand-alone; names and signatures follow the in-tree call sites
// (/root/reference src/common/Module/Filter/Filter.hxx:56-96, Framer/Framer.hxx:53-76,
//  src/mains/TX_RX_BB/main.cpp:75-96).  With a real StreamPU, include <streampu.hpp> instead.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <typeindex>
#include <vector>

namespace spu {
namespace tools {
class exception : public std::runtime_error {
public:
    exception(const std::string &file, int line, const std::string &func, const std::string &msg)
        : std::runtime_error(file + ":" + std::to_string(line) + " (" + func + "): " + msg) {}
};
#define SPU_DECL_EXC(name) \
    class name : public exception { public: using exception::exception; \
        name(const std::string &f, int l, const std::string &fn) : exception(f, l, fn, #name) {} };
SPU_DECL_EXC(invalid_argument)
SPU_DECL_EXC(length_error)
SPU_DECL_EXC(runtime_error)
SPU_DECL_EXC(unimplemented_error)
SPU_DECL_EXC(cannot_allocate)
SPU_DECL_EXC(processing_aborted)
#undef SPU_DECL_EXC
}  // namespace tools

namespace module {
class Module;
}
namespace runtime {
class Task;
class Socket {
public:
    enum class dir { in, out };
    Socket(Task &t, std::string name, std::type_index ty, size_t elt_size, size_t n_elmts, dir d)
        : task(t), name_(std::move(name)), type(ty), elt_size(elt_size), n_elmts_(n_elmts), dir_(d) {}
    const std::string &get_name() const { return name_; }
    size_t get_n_elmts() const { return n_elmts_; }          // n_frames * per-frame elements
    size_t get_databytes() const { return n_elmts_ * elt_size; }
    void *get_dataptr() const { return ptr; }
    template <typename T> T *get_dataptr() const { return static_cast<T *>(ptr); }
    bool is_out() const { return dir_ == dir::out; }
    // binding: `in_socket = out_socket` makes the input read the producer's buffer (main.cpp:75-94)
    void operator=(Socket &producer)
    {
        if (dir_ != dir::in || !producer.is_out())
            throw tools::invalid_argument(__FILE__, __LINE__, __func__, "bind an input socket to an output socket");
        if (producer.type != type || producer.get_databytes() != get_databytes()) {
            std::stringstream m;
            m << "socket '" << producer.name_ << "' (" << producer.get_databytes() << " B) cannot feed '" << name_ << "' ("
              << get_databytes() << " B)";
            throw tools::length_error(__FILE__, __LINE__, __func__, m.str());
        }
        ptr = producer.ptr;
        bound_from = &producer;
    }
    // a raw std::vector can be bound as a constant input (sigma, main.cpp:74,82)
    template <typename T, typename A> void operator=(std::vector<T, A> &v)
    {
        if (std::type_index(typeid(T)) != type || v.size() * sizeof(T) != get_databytes())
            throw tools::length_error(__FILE__, __LINE__, __func__, "vector size does not match socket '" + name_ + "'");
        ptr = v.data();
    }
    Task &task;
    Socket *bound_from = nullptr;

private:
    friend class Task;
    friend class spu::module::Module;
    std::string name_;
    std::type_index type;
    size_t elt_size, n_elmts_;
    dir dir_;
    void *ptr = nullptr;
    std::vector<uint8_t> storage;     // output sockets own their buffer
};
}  // namespace runtime

namespace module {
class Module;
}
namespace runtime {
class Task {
public:
    Task(module::Module &m, std::string name) : module(m), name_(std::move(name)) {}
    const std::string &get_name() const { return name_; }
    Socket &operator[](size_t i) { return *sockets.at(i); }
    int exec();
    void set_debug(bool) {}
    void set_stats(bool) {}
    void set_fast(bool) {}
    std::vector<std::unique_ptr<Socket>> sockets;
    std::function<int(module::Module &, Task &, size_t)> codelet;
    module::Module &module;
    uint64_t n_calls = 0;

private:
    std::string name_;
};
}  // namespace runtime

namespace module {
class Module {
public:
    virtual ~Module() = default;
    void set_name(const std::string &n) { name = n; }
    void set_short_name(const std::string &n) { short_name = n; }
    const std::string &get_name() const { return name; }
    virtual void set_n_frames(size_t n)
    {
        if (n == 0) throw tools::invalid_argument(__FILE__, __LINE__, __func__, "'n_frames' has to be greater than 0");
        n_frames = n;
    }
    size_t get_n_frames() const { return n_frames; }
    runtime::Task &operator[](size_t t) { return *tasks.at(t); }
    std::vector<std::unique_ptr<runtime::Task>> tasks;

protected:
    runtime::Task &create_task(const std::string &n)
    {
        tasks.emplace_back(new runtime::Task(*this, n));
        return *tasks.back();
    }
    template <typename T> size_t create_socket_in(runtime::Task &t, const std::string &n, size_t n_elmts)
    {
        t.sockets.emplace_back(new runtime::Socket(t, n, typeid(T), sizeof(T), n_elmts * n_frames, runtime::Socket::dir::in));
        return t.sockets.size() - 1;
    }
    template <typename T> size_t create_socket_out(runtime::Task &t, const std::string &n, size_t n_elmts)
    {
        auto *s = new runtime::Socket(t, n, typeid(T), sizeof(T), n_elmts * n_frames, runtime::Socket::dir::out);
        s->storage.resize(s->get_databytes());
        s->ptr = s->storage.data();
        t.sockets.emplace_back(s);
        return t.sockets.size() - 1;
    }
    void create_codelet(runtime::Task &t, std::function<int(Module &, runtime::Task &, size_t)> c) { t.codelet = std::move(c); }
    std::string name, short_name;
    size_t n_frames = 1;
};
using Stateful = Module;
}  // namespace module

inline int runtime::Task::exec()
{
    for (auto &s : sockets)
        if (!s->get_dataptr())
            throw tools::runtime_error(__FILE__, __LINE__, __func__, "socket '" + s->get_name() + "' of task '" + name_ + "' is not bound");
    n_calls++;
    return codelet(module, *this, (size_t)-1);
}

namespace runtime {
// Tasks run in the given order, once per exec() iteration, until stop() is true.
class Sequence {
public:
    explicit Sequence(std::vector<Task *> order) : order(std::move(order)) {}
    void exec(const std::function<bool()> &stop)
    {
        do {
            for (Task *t : order) t->exec();
        } while (!stop());
    }
    void exec_step() { for (Task *t : order) t->exec(); }

private:
    std::vector<Task *> order;
};
}  // namespace runtime
}  // namespace spu
