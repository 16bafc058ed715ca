// stage3.hpp - engine 3 (persistent stage segments); see DESIGN.md
