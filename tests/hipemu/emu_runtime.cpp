// storage of the emulator's per-thread launch context (see include/hip/hip_runtime.h)
#include <hip/hip_runtime.h>
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
namespace hipemu {
thread_local Launch* cur = nullptr;
thread_local Wave* wave = nullptr;
thread_local int tid_flat = 0;
}  // namespace hipemu


#ifndef HIPEMU_UCONTEXT
// void hipemu_switch(void** save_sp, void* load_sp): park the running lane (callee-saved registers on its stack, stack
// pointer into *save_sp) and resume the one whose stack pointer is load_sp.  System V x86-64; no signal mask, no FPU
// control state (the kernels never change it).
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch,.-hipemu_switch
.section .note.GNU-stack,"",@progbits
)");
#endif
